#!/usr/bin/env python3
"""bench.py -- headline measurement of the fake-quantization hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch of synthetic input: one call of
quantization.uniformQuantization(x, s=16, bucket_size=256) on a 64 Mi-element fp32 tensor
(BASELINE.json: "quantize-dequantize GB/s (64M fp32, 4-bit)"), through the public Python API
(allocation of the result + one HIP kernel launch through the C ABI).  Inputs are resident in
HBM; four input tensors and four live outputs are rotated so that every call streams 512 MiB
through HBM and cannot be served from the 256 MiB Infinity Cache.

Multi-GPU (one process per GPU): the path shards by tensor -- every rank quantizes its own
tensors, no collective in the data path -- so scaling is "weak" and value = total bytes of all
ranks / max-over-ranks time.  `python bench.py --gpus N` started WITHOUT a torchrun environment
launches the N ranks itself (harness/launch.py re-executes this file under
torch.distributed.run, 127.0.0.1 rendezvous, backend "nccl" = RCCL); started under torchrun it
uses the ranks it was given.  At N=1 a single-rank RCCL group is still created, so the gradient
all-reduce of the steps/sec legs really runs through RCCL on a one-GPU box.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      achieved algorithmic GB/s of the dominant kernel (8 B/element: 4 read + 4 written,
                SURVEY.md 8d) from HIP-event timing of the timed region, against the 8 TB/s peak;
                roofline.kernels: the same figure for every other kernel on the path (SURVEY 8d's
                secondary rows, K2 ... K9, the multi-tensor kernels on the WRN-16-22 shape list),
                HIP-event timed in this run by harness/kernel_bench.py.  The driver's record keeps
                only the SCALARS of this object, so every row is also there as a short string
                (k01, k02, ...), and so are the steps/sec and data-parallel figures of the
                `distill` object (steps_cfg*, dp_cfg*): mirrors, the full objects stay in the line
  cpu_baseline  the REFERENCE's own uniformQuantization (staged bytecode of
                /root/reference/quantization, oracle/ref_stage.py) timed on the host cores of this
                box on the same workload, with the two ports (C/OpenMP, torch ops) next to it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_ELEM = 64 * 1024 * 1024
LEVELS = 16
BUCKET = 256
ALGO_BYTES_PER_ELEM = 8            # 4 B read + 4 B written (alpha/beta side outputs: 0.03 B/elem, not counted)
HBM_PEAK_GBPS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_ROTATE = 4


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _time_runs(fn, min_runs, budget_s):
    """Warm-up + >= min_runs timed runs (more while the time budget lasts, at most 10)."""
    fn()
    ts = []
    t_end = time.time() + budget_s
    while len(ts) < min_runs or (len(ts) < 10 and time.time() < t_end):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def cpu_baseline(x_host, q_gpu, alpha_gpu):
    """The reference's own quantizer on the host cores of this box, same workload (bounded sample);
    its output doubles as the checker of the GPU result for the same tensor (q_gpu, alpha_gpu).
    ref: quantization/quant_functions.py:155-194."""
    import numpy as np
    from oracle import oracle_c, ref_stage
    from oracle.torch_port import uniform_quantize_torch_ops
    xn = x_host.numpy()
    n = xn.size
    ncpu = os.cpu_count() or 1
    out = {'unit': 'GB/s', 'cpu_model': cpu_model(), 'os_cpu_count': ncpu}

    refq = ref_stage.load()
    if refq is not None:
        # torch's elementwise CPU ops oversubscribe badly with one thread per SMT sibling on a 2-socket box
        # (0.3 GB/s at 256 threads in round 1), so the reference is timed at several thread counts and the BEST is
        # the baseline; os.cpu_count() threads -- what the survey prescribes -- is always among them.
        counts = sorted({ncpu, max(1, ncpu // 2), min(ncpu, 64), min(ncpu, 32)}, reverse=True)
        per_threads, best = {}, None
        for th in counts:
            torch.set_num_threads(th)
            ts = _time_runs(lambda: refq.uniformQuantization(x_host, LEVELS, bucket_size=BUCKET), 5, 4.0)
            per_threads[str(th)] = {'min_s': round(min(ts), 4), 'median_s': round(float(np.median(ts)), 4), 'runs': len(ts),
                                    'GBps_at_min': round(ALGO_BYTES_PER_ELEM * n / min(ts) / 1e9, 3)}
            if best is None or min(ts) < best[1]:
                best = (th, min(ts), float(np.median(ts)), len(ts))
        torch.set_num_threads(best[0])
        q_ref, sf_ref = refq.uniformQuantization(x_host, LEVELS, bucket_size=BUCKET)
        out.update({
            'value': round(ALGO_BYTES_PER_ELEM * n / best[1] / 1e9, 3), 'cores': best[0], 'kind': 'reference',
            'sample': "%d runs after 1 warm-up of the full workload (N=%d fp32, s=%d, bucket=%d) with the reference's own "
                      'quantization.uniformQuantization (bytecode of /root/reference/quantization staged by oracle/ref_stage.py), '
                      'torch %s CPU ops, torch.set_num_threads(%d) = best of the thread counts tried; min %.4f s, median %.4f s'
                      % (best[3], n, LEVELS, BUCKET, torch.__version__, best[0], best[1], best[2]),
            'threads_tried': per_threads,
            'reference_sources_sha256': (ref_stage.manifest() or {}).get('files'),
        })
        bit_exact = bool(np.array_equal(q_gpu, q_ref.numpy()) and
                         np.array_equal(alpha_gpu, sf_ref.alpha.numpy().reshape(-1)))
        out['gpu_result_bit_exact_vs_reference'] = bit_exact
        del q_ref, sf_ref
    else:
        out['reference_error'] = ('oracle/_ref is not staged (run __graft_entry__.build() where /root/reference exists); '
                                  'falling back to the C port as the baseline')

    # secondary: the two ports of the same algorithm (test infrastructure, oracle/)
    oracle_c.build()
    cores = oracle_c.max_threads()
    ref = oracle_c.uniform_quantize(xn, LEVELS, BUCKET, want_idx=False, want_lev=False)       # warm-up + checker
    out['gpu_result_bit_exact'] = bool(np.array_equal(q_gpu, ref['q']) and np.array_equal(alpha_gpu, ref['alpha']))
    del ref
    ts = _time_runs(lambda: oracle_c.uniform_quantize(xn, LEVELS, BUCKET, want_idx=False, want_lev=False), 3, 6.0)
    out['c_port'] = {'value': round(ALGO_BYTES_PER_ELEM * n / min(ts) / 1e9, 3), 'unit': 'GB/s', 'threads': cores,
                     'sample': '%d runs, min %.4f s, median %.4f s; oracle/qd_oracle.c, OpenMP over buckets'
                               % (len(ts), min(ts), float(np.median(ts)))}
    if 'value' not in out:
        out.update({'value': out['c_port']['value'], 'cores': cores, 'kind': 'port', 'sample': out['c_port']['sample']})
    torch.set_num_threads(min(ncpu, 64))
    tt = _time_runs(lambda: uniform_quantize_torch_ops(x_host, LEVELS, BUCKET), 3, 6.0)
    out['torch_ops_port'] = {
        'value': round(ALGO_BYTES_PER_ELEM * n / min(tt) / 1e9, 3), 'unit': 'GB/s', 'threads': torch.get_num_threads(),
        'sample': '%d runs, min %.4f s, median %.4f s; same sequence of torch CPU ops as '
                  'quantization/quant_functions.py:155-194 (oracle/torch_port.py)' % (len(tt), min(tt), float(np.median(tt))),
    }
    out['distill'] = cpu_distill_baseline(refq)
    return out


def cpu_distill_baseline(refq=None, steps=200, warmup=3, batch=50):
    """BASELINE configs[0]: the CIFAR10 ConvolForwardNet student step on the CPU with the
    reference's own quantizer (staged bytecode; its torch-op port when nothing is staged) in the
    reference's loop shape (quantize every parameter, fwd/bwd with the KD loss, restore, SGD): the
    200 steps of configs[0]'s "1 epoch synthetic" (10000 images / batch 50), ~25 s."""
    from harness import models
    from oracle.torch_port import uniform_quantize_torch_ops
    if refq is not None:
        def quantize_one(t):
            return refq.uniformQuantization(t, 16, bucket_size=256)[0]
    else:
        def quantize_one(t):
            return uniform_quantize_torch_ops(t, 16, 256)[0]
    torch.manual_seed(0)
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    st, te = models.student().train(), models.teacher().eval()
    opt = torch.optim.SGD(st.parameters(), lr=1e-3, momentum=0.9, nesterov=True, weight_decay=2.2e-4)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(batch, 3, 32, 32, generator=g), torch.randint(0, 10, (batch,), generator=g)
    t_quant = 0.0

    def one():
        nonlocal t_quant
        a = time.perf_counter()
        saved = [p.data for p in st.parameters()]
        for p in st.parameters():
            p.data = quantize_one(p.data)
        t_quant += time.perf_counter() - a
        opt.zero_grad()
        with torch.no_grad():
            t_out = te(x)
        models.kd_loss(st(x), t_out, y).backward()
        for p, m in zip(st.parameters(), saved):
            p.data = m
        opt.step()

    for _ in range(warmup):
        one()
    t_quant = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {'steps_per_sec': round(steps / dt, 3), 'ms_per_step': round(dt / steps * 1e3, 2),
            'quantize_ms_per_step': round(t_quant / steps * 1e3, 3), 'threads': threads,
            'quantizer': 'reference (oracle/_ref bytecode)' if refq is not None else 'torch-op port (oracle/torch_port.py)',
            'sample': '%d steps (one synthetic epoch: 10000 images) after %d warm-up steps, batch %d, synthetic CIFAR10-shaped '
                      'data; student+teacher fwd, KD loss, bwd, SGD on the host with the reference quantizer in the loop '
                      '(configs[0])' % (steps, warmup, batch)}


from harness.dpbench import XGMI_PEAK_GBPS, dp_report, event_ms, flat_dp, timed_steps  # noqa: E402,F401


def distill_steps_per_sec(dev, rank, n_gpus, distributed, ctl_barrier, steps=100, warmup=20, per_gpu_batch=50, repetitions=7):
    """Second half of BASELINE.json's metric: distilled-training steps/sec on synthetic
    CIFAR10-shaped data (configs[1]: ConvolForwardNet student, 4-bit uniform quantization, bucket
    256, pure STE), data parallel over the ranks with one RCCL all-reduce of the flat gradient
    per step.  Weak scaling: per-GPU batch fixed at 50."""
    import torch.distributed as dist
    from harness import models
    from harness.distill import DistillTrainer, synthetic_batch
    torch.manual_seed(0)                                   # identical replicas on every rank
    out = {'config': 'CIFAR10-shaped synthetic randn(B,3,32,32), ConvolForwardNet student (22 tensors, 1.00 M '
                     'params) distilled from the 5.3 M teacher, KD loss T=2, SGD nesterov, 4-bit uniform, bucket 256, STE',
           'per_gpu_batch': per_gpu_batch, 'global_batch': per_gpu_batch * n_gpus, 'steps': steps, 'warmup': warmup}
    import statistics
    modes = ('multi', 'per_tensor')
    batches = [synthetic_batch(per_gpu_batch, dev, seed=1000 * rank + i) for i in range(4)]
    trainers = {}
    for mode in modes:
        torch.manual_seed(0)
        trainers[mode] = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode=mode)
        for i in range(warmup):
            trainers[mode].step(*batches[i % 4])
    # third leg: the multi-tensor step replayed from hipGraphs (quantize + forward + loss + backward in one graph, the
    # optimizer in a second, the RCCL all-reduce eager between the two): the ~150 launches of a step stop depending on the
    # host, which is where the repetition-to-repetition spread of the eager legs comes from
    graph_error = None
    try:
        torch.manual_seed(0)
        tg = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi')
        for i in range(warmup):
            tg.step(*batches[i % 4])
        tg.capture(*batches[0])                             # local: no collective inside
    except Exception as e:                                  # noqa: BLE001 -- reported in the JSON, the eager legs still run
        graph_error = '%s: %s' % (type(e).__name__, e)
    if distributed:
        # every rank must run the same legs (each holds collectives): the leg is dropped everywhere if the capture failed anywhere
        ok = torch.tensor([0 if graph_error else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            graph_error = graph_error or 'graph capture failed on another rank'
    if graph_error is None:
        for i in range(warmup):
            tg.step(*batches[i % 4])
        trainers['multi_graph'] = tg
        modes = modes + ('multi_graph',)

    # The 2 ms step is ~150 small launches (MIOpen's small-shape convolutions, batch-norm, the optimizer) and its time
    # moves from repetition to repetition on one box (docs/history/profiles/r03_distill_spread.txt).  So: REPS repetitions of `steps`
    # steps per mode, INTERLEAVED (multi, per_tensor, multi, ...) so that drift hits both alike; the MEDIAN is reported,
    # every repetition is listed, and the two modes are only called different when their ranges do not overlap.
    REPS = repetitions
    reps = {m: [] for m in modes}
    for _rep in range(REPS):
        for mode in modes:
            tr = trainers[mode]
            job, _own = timed_steps(lambda i, tr=tr: tr.step(*batches[i % 4]), steps, 1, dev, distributed)
            reps[mode].append(job[0])
    for mode in modes:
        tr = trainers[mode]
        dt = statistics.median(reps[mode])
        sps = sorted(steps / r for r in reps[mode])
        out[mode] = {'steps_per_sec': round(steps / dt, 2), 'ms_per_step': round(dt / steps * 1e3, 4),
                     'samples_per_sec': round(steps * per_gpu_batch * n_gpus / dt, 1), 'statistic': 'median of %d repetitions' % REPS,
                     'steps_per_sec_min': round(sps[0], 1), 'steps_per_sec_max': round(sps[-1], 1),
                     'steps_per_sec_repetitions': [round(steps / r, 1) for r in reps[mode]]}
        if mode == 'multi_graph':
            continue
        # per-phase breakdown, each phase HIP-event timed on its own over >= 50 back-to-back calls (serialised, so the sum
        # exceeds the step)
        x, y = batches[0]
        out[mode]['phases'] = {
            'quantize_ms': round(event_ms(tr.quantize, 50), 4),
            'fwd_bwd_ms': round(event_ms(lambda: tr.forward_backward(x, y), 20, precondition_s=0.05), 4),
            'restore_ms': round(event_ms(tr.restore, 50, precondition_s=0.02), 4),
            'allreduce_ms': round(event_ms(tr.sync.sync, 50, precondition_s=0.02), 4),
            'optimizer_ms': round(event_ms(tr.opt.step, 50, precondition_s=0.02), 4),
            'timing': 'HIP events, median of 3 repetitions of 20-50 calls after preconditioning'}
    lo_m, hi_m = out['multi']['steps_per_sec_min'], out['multi']['steps_per_sec_max']
    lo_p, hi_p = out['per_tensor']['steps_per_sec_min'], out['per_tensor']['steps_per_sec_max']
    out['multi_vs_per_tensor'] = ('multi faster in every repetition' if lo_m > hi_p else
                                  'per_tensor faster in every repetition' if lo_p > hi_m else
                                  'indistinguishable: the repetition ranges overlap (the quantizer is %.3f / %.3f ms of the step)'
                                  % (out['multi']['phases']['quantize_ms'], out['per_tensor']['phases']['quantize_ms']))
    if 'multi_graph' in out:
        lo_g, hi_g = out['multi_graph']['steps_per_sec_min'], out['multi_graph']['steps_per_sec_max']
        out['multi_graph_vs_multi'] = ('graph replay faster in every repetition' if lo_g > hi_m else
                                       'eager faster in every repetition' if lo_m > hi_g else
                                       'indistinguishable: the repetition ranges overlap')
    # the data-parallel figures of this config, on the graph-replay trainer when there is one (its all-reduce is the eager
    # call between the two graphs), else on the eager multi-tensor one
    best = 'multi_graph' if 'multi_graph' in trainers else 'multi'
    tr = trainers[best]

    def set_exchange(on, tr=tr):
        tr.sync.active = on and tr.sync.world_active
    out['dp'] = dp_report(lambda i: tr.step(*batches[i % 4]), steps, 3, dev, n_gpus, distributed, per_gpu_batch,
                          tr.flat_grad.numel() * 4, set_exchange, tr.sync.sync if tr.sync.active else None, ctl_barrier, rank)
    out['dp']['trainer'] = best
    trainers.clear()
    if graph_error is not None:
        out['multi_graph'] = {'error': graph_error}
    out['note'] = ("'multi' = one multi-tensor quantize launch per step on persistent shadows (K9); 'per_tensor' = the "
                   "reference's loop shape (22 uniformQuantization calls + restore); 'multi_graph' = the 'multi' step replayed "
                   "from two hipGraphs (DistillTrainer.capture; tests/test_hip_distill.py::test_graph_replay_matches_eager): the "
                   "GPU work is the same MIOpen small-shape kernels, but ~150 launches per step no longer wait for the host, so "
                   "the repetitions stop spreading")
    return out


def dp_config_steps_per_sec(kind, dev, rank, n_gpus, distributed, ctl_barrier, steps=10, warmup=3, reps=3):
    """BASELINE configs[3] (kind='imagenet': ImageNet-shaped synthetic, resnet_kfilters
    resnet18(k=1.5) student distilled from a ResNet-34-shaped teacher, 4-bit bucketed, first/last
    tensors not quantized, DP over 8 GPUs) and configs[4] (kind='nmt': 2-layer LSTM seq2seq,
    multi30k-shaped synthetic tokens, 4-bit quantized distillation, DP over 4 GPUs).  Data
    parallel with the flat-gradient RCCL all-reduce, cut in 4 pieces overlapped with backward."""
    from harness import models
    from harness.distill import (DistillTrainer, seq2seq_kd_loss_fn, synthetic_batch, synthetic_token_batch)
    torch.manual_seed(0)
    if kind == 'imagenet':
        per_gpu = 32
        tr = DistillTrainer(models.ResNetK((2, 2, 2, 2), 1.5), models.ResNetK((3, 4, 6, 3), 1.0), dev, num_bits=4,
                            bucket_size=256, lr=0.1, weight_decay=1e-4, quantize_first_and_last_layer=False,
                            grad_chunks=4, overlap_allreduce=True)
        batches = [synthetic_batch(per_gpu, dev, seed=1000 * rank + i, classes=1000, side=224) for i in range(2)]
        desc = ('ImageNet-shaped synthetic randn(B,3,224,224), 1000 classes; resnet18(k=1.5) student (62 tensors, 25.9 M) '
                'distilled from a ResNet-34-shaped teacher; SGD nesterov lr 0.1 wd 1e-4; 4-bit uniform, bucket 256, '
                'quantize_first_and_last_layer=False')
    else:
        per_gpu = 64
        tr = DistillTrainer(models.Seq2SeqLSTM(), models.Seq2SeqLSTM(), dev, num_bits=4, bucket_size=256, lr=1.0,
                            momentum=0.0, nesterov=False, weight_decay=0.0, loss_fn=seq2seq_kd_loss_fn, clip_norm=5.0,
                            grad_chunks=4, overlap_allreduce=True,
                            quantize_from_first_step=False)      # ref: translation_models/model.py:184,243
        batches = [synthetic_token_batch(per_gpu, dev, seed=1000 * rank + i) for i in range(2)]
        desc = ('multi30k-shaped synthetic tokens (len 20..50, V_src 18000, V_tgt 10000), 2-layer LSTM 500/500 with input '
                'feeding + general attention (22 tensors, 28.8 M), teacher of the same shape, word-level KD 0.3 NLL + 0.7 KL; '
                'SGD lr 1.0, clip-norm 5; 4-bit uniform, bucket 256')
    for i in range(warmup):
        tr.step(*batches[i % 2])

    def set_exchange(on):
        tr.sync.active = on and tr.sync.world_active
    out = {'config': desc}
    out.update(dp_report(lambda i: tr.step(*batches[i % 2]), steps, reps, dev, n_gpus, distributed, per_gpu,
                         tr.flat_grad.numel() * 4, set_exchange, tr.sync.sync if tr.sync.active else None, ctl_barrier, rank))
    out['gradient_bytes_per_step'] = int(tr.flat_grad.numel() * 4)
    out['allreduce_shape'] = ('%d asynchronous RCCL all-reduces (ReduceOp.AVG) of ~equal bytes, launched from the backward hooks in '
                              'gradient-arrival order' % len(tr.sync.bounds)) if tr.sync.active else 'none (one rank, not forced)'
    out['phases'] = {
        'quantize_ms': round(event_ms(tr.quantize, 50), 4),
        'fwd_bwd_ms (+overlapped all-reduce launch)': round(event_ms(lambda: (tr.forward_backward(*batches[0]), tr.sync.sync()), 5, precondition_s=0.0, reps=2), 3),
        'optimizer_ms': round(event_ms(tr.opt.step, 20, precondition_s=0.02), 4),
        'timing': 'HIP events; quantize: median of 3 x 50 launches after 100 ms of preconditioning'}
    nq = sum(m.numel() for m, q in zip(tr.masters, tr.quantized) if q)
    out['phases']['quantize_GBps'] = round(8 * nq / (out['phases']['quantize_ms'] * 1e-3) / 1e9, 1)
    out['phases']['quantize_frac_of_8TBps'] = round(out['phases']['quantize_GBps'] / HBM_PEAK_GBPS, 4)
    del tr
    torch.cuda.empty_cache()
    return out


def diffquant_steps_per_sec(dev, rank, n_gpus, distributed, ctl_barrier, steps=8, warmup=2, batch=100, reps=3):
    """BASELINE configs[2]: CIFAR10 WideResNet-16-22 student (60 tensors, 82.7 M parameters), 2-bit
    (k = 4 points) non-uniform differentiable quantization, bucket 256: steps/sec of the
    optimize_quantization_points loop with the per-step quantizer cost broken out.  Quoted on 1 GPU;
    at N > 1 it runs data parallel, exchanging only the ntensors x k point gradients."""
    import torch.distributed as dist
    from harness import models
    from harness.diffquant import DiffQuantTrainer
    from harness.distill import synthetic_batch
    torch.manual_seed(0)
    t0 = time.perf_counter()
    tr = DiffQuantTrainer(models.WideResNet(16, 22), dev, num_points=4, bucket_size=256, lr=1e-5, mode='multi')
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    x, y = synthetic_batch(batch, dev, seed=11 + 1000 * rank)
    for _ in range(warmup):
        tr.step(x, y)
    exchanging = tr.exchange

    def set_exchange(on):
        tr.exchange = on and exchanging

    def exchange_once():
        dist.all_reduce(tr.points_grad)
    nparams = sum(p.numel() for p in tr.params)
    out = {'config': 'Wide_ResNet depth 16 widen 22 (60 tensors, %.1f M params), k=4 points (2-bit) per tensor, bucket 256, '
                     'percentile init, KD loss vs the unquantized model, SGD on the points; batch %d synthetic CIFAR10-shaped'
                     % (nparams / 1e6, batch)}
    out.update(dp_report(lambda i: tr.step(x, y), steps, reps, dev, n_gpus, distributed, batch,
                         tr.points_grad.numel() * 4 if exchanging else 0, set_exchange, exchange_once if exchanging else None,
                         ctl_barrier, rank))
    out['setup_s'] = round(setup_s, 2)
    ph = {'assign_all_tensors_ms (multi-tensor K5, 1 launch)': round(event_ms(tr.quantize, 50), 4),
          'fwd_bwd_ms': round(event_ms(lambda: tr.forward_backward(x, y), 3, precondition_s=0.0, reps=2), 3),
          'point_gradients_ms (multi-tensor K6, 2 launches)': round(event_ms(tr.point_gradients, 50), 4),
          'timing': 'HIP events; K5m / K6m: median of 3 x 50 launches after 100 ms of preconditioning'}
    nq = sum(tr.params[i].numel() for i in tr.slots)
    ph['assign_GBps (9 B/elem)'] = round(9 * nq / (ph['assign_all_tensors_ms (multi-tensor K5, 1 launch)'] * 1e-3) / 1e9, 1)
    ph['point_gradients_GBps (5 B/elem)'] = round(5 * nq / (ph['point_gradients_ms (multi-tensor K6, 2 launches)'] * 1e-3) / 1e9, 1)
    out['phases'] = ph
    out['reference_cpu_quantizer_note'] = ('reference per-step quantizer cost on this model, CPU path: ~2.5 s per 16 Mi-element '
                                           'tensor (BASELINE.md section 3); here the 60-tensor assign + point-gradient pair is the '
                                           'two phase entries above')
    return out


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if present."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get('k_bucket_vec_hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def pmc_slice(launches=12):
    """Child of measure_pmc_traffic() / measure_rocprof_duration(): `launches` launches of the headline call under rocprofv3,
    nothing else."""
    import quantization
    dev = torch.device('cuda', 0)
    gen = torch.Generator().manual_seed(0)
    nbuf = 2 if launches <= 12 else N_ROTATE
    xs = [torch.randn(N_ELEM, generator=gen).to(dev) for _ in range(nbuf)]
    live = [None] * nbuf
    for i in range(launches):
        live[i % nbuf] = quantization.uniformQuantization(xs[i % nbuf], LEVELS, bucket_size=BUCKET)[0]
    torch.cuda.synchronize()
    if launches <= 12:
        # the PMC passes also see a few launches of the other per-step kernels (OTHER_PMC_KERNELS): 6 each, N = 64 Mi
        sf = quantization.ScalingFunction('linear', False, False, BUCKET)
        for i in range(6):
            live[i % nbuf] = sf.scale_down(xs[i % nbuf])                                               # K2
        pts = torch.tensor([0.0, 0.3, 0.7, 1.0], device=dev)
        fns = [quantization.nonUniformQuantization_variable(bucket_size=BUCKET, pre_process_tensors=True, tensor=x) for x in xs]
        g = torch.randn(N_ELEM, generator=gen).to(dev)
        for i in range(6):
            fns[i % nbuf].forward(None, pts)                                                           # K5
        for i in range(6):
            fns[i % nbuf].backward(g)                                                                  # K6
        torch.cuda.synchronize()


# kernel-name substring -> (label, algorithmic bytes per launch) of what pmc_slice() launches beside the headline kernel;
# the K2 entry also matches the two scale_down launches nonUniformQuantization_variable's constructor makes (same bytes)
OTHER_PMC_KERNELS = {
    'k_bucket_vec<1, 16, 4, 1>': ('K2 scale_down', 8 * N_ELEM),
    'k_nearest_prescaled_stream<false>': ('K5 diff-quant forward', 9 * N_ELEM),
    'k_point_grad_fast<4, 1, 1': ('K6 point gradient', 5 * N_ELEM),
}


def measure_rocprof_duration(launches=1200, timeout_s=150):
    """The headline kernel's average duration as rocprofv3 sees it, IN THIS RUN: `rocprofv3 --kernel-trace --stats` over a child
    process that does `launches` back-to-back launches of the same call (the first third is dropped: clocks and allocator
    settle).  What roofline.avg_launch_us (HIP events around the timed region, gaps included) has to agree with."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return {'error': 'rocprofv3 not found'}
    with tempfile.TemporaryDirectory(dir='/tmp') as td:
        env = dict(os.environ, TMPDIR='/tmp')
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'QD_FORCE_DIST'):
            env.pop(k, None)
        cmd = [exe, '--kernel-trace', '--output-format', 'csv', '-d', td, '-o', 'dur', '--',
               sys.executable, os.path.abspath(__file__), '--pmc-slice', '--pmc-slice-launches', str(launches)]
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return {'error': 'rocprofv3 --kernel-trace timed out after %d s' % timeout_s}
        files = glob.glob(os.path.join(td, '**', '*kernel_trace.csv'), recursive=True)
        if r.returncode != 0 or not files:
            return {'error': 'rocprofv3 --kernel-trace: rc %d, %d trace files; %s' % (r.returncode, len(files), r.stderr.decode(errors='replace')[-300:])}
        d = []
        with open(files[0]) as fh:
            for row in csv.DictReader(fh):
                if 'k_bucket_vec' in row.get('Kernel_Name', ''):
                    d.append((int(row['Start_Timestamp']), (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3))
    d = [us for _t, us in sorted(d)][len(d) // 3:]
    if not d:
        return {'error': 'no k_bucket_vec dispatch in the trace'}
    return {'avg_us': round(sum(d) / len(d), 3), 'min_us': round(min(d), 3), 'max_us': round(max(d), 3), 'launches': len(d),
            'how': 'rocprofv3 --kernel-trace over %d launches of the headline call in a child process of this run; per-dispatch '
                   'End - Start of k_bucket_vec, the first third dropped' % launches}


def measure_pmc_traffic(timeout_s=150):
    """HBM bytes per launch of the headline kernel MEASURED IN THIS RUN: two short rocprofv3 passes (--pmc FETCH_SIZE, then
    --pmc WRITE_SIZE, each with --kernel-trace only, as MI355X_MICROARCH.md prescribes: the two counters do not fit one
    pass) over a 12-launch slice of the same call in a child process, after the timed region.  FETCH_SIZE / WRITE_SIZE
    are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read, so read bytes = 2 x FETCH_SIZE x 1024
    (the guide's correction).  Returns a dict; on any failure {'error': ...} -- the headline number never depends on it."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return {'error': 'rocprofv3 not found'}
    raw, other_raw = {}, {}
    t_start = time.time()
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        with tempfile.TemporaryDirectory(dir='/tmp') as td:
            env = dict(os.environ, TMPDIR='/tmp')
            for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'QD_FORCE_DIST'):
                env.pop(k, None)
            cmd = [exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', td, '-o', 'pmc', '--',
                   sys.executable, os.path.abspath(__file__), '--pmc-slice']
            try:
                r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return {'error': 'rocprofv3 --pmc %s timed out after %d s' % (counter, timeout_s)}
            files = glob.glob(os.path.join(td, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return {'error': 'rocprofv3 --pmc %s: rc %d, %d counter files; %s'
                                 % (counter, r.returncode, len(files), r.stderr.decode(errors='replace')[-300:])}
            vals, others = {}, {k: {} for k in OTHER_PMC_KERNELS}
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    name = row.get('Kernel_Name', '')
                    if row.get('Counter_Name') != counter:
                        continue
                    if 'k_bucket_vec<0, 16, 4, 1>' in name:
                        vals[int(row['Dispatch_Id'])] = vals.get(int(row['Dispatch_Id']), 0.0) + float(row['Counter_Value'])
                    for sub in OTHER_PMC_KERNELS:
                        if sub in name:
                            d_ = others[sub]
                            d_[int(row['Dispatch_Id'])] = d_.get(int(row['Dispatch_Id']), 0.0) + float(row['Counter_Value'])
            v = [vals[k] for k in sorted(vals)][2:]                       # drop the first two launches
            if not v:
                return {'error': 'no k_bucket_vec dispatch in the %s pass' % counter}
            raw[counter] = {'per_launch_KiB_avg': sum(v) / len(v), 'launches': len(v), 'min': min(v), 'max': max(v)}
            for sub, d_ in others.items():
                w = [d_[k] for k in sorted(d_)][1:]
                if w:
                    other_raw.setdefault(sub, {})[counter] = sum(w) / len(w)
    read_b = 2.0 * raw['FETCH_SIZE']['per_launch_KiB_avg'] * 1024
    write_b = raw['WRITE_SIZE']['per_launch_KiB_avg'] * 1024
    algo = ALGO_BYTES_PER_ELEM * N_ELEM
    other = {}
    for sub, (label, abytes) in OTHER_PMC_KERNELS.items():
        c = other_raw.get(sub, {})
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            hb = 2.0 * c['FETCH_SIZE'] * 1024 + c['WRITE_SIZE'] * 1024
            other[label] = {'bytes_per_launch': round(hb), 'over_algorithmic': round(hb / abytes, 4), 'algorithmic_bytes_per_launch': abytes}
    return {'bytes_per_launch': round(read_b + write_b), 'read_bytes_per_launch': round(read_b), 'write_bytes_per_launch': round(write_b),
            'over_algorithmic': round((read_b + write_b) / algo, 4), 'raw_KiB': raw, 'other_kernels': other, 'seconds': round(time.time() - t_start, 1),
            'how': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace over 12 launches of the headline call in '
                   'a child process of this run; read = 2 x FETCH_SIZE x 1024 (gfx950 halves wide streaming reads), '
                   'write = WRITE_SIZE x 1024'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernels', action='store_true', help='skip the per-kernel roofline rows (roofline.kernels)')
    ap.add_argument('--no-distill', action='store_true', help='skip the distilled-training steps/sec leg')
    ap.add_argument('--no-diffquant', action='store_true', help='skip the WideResNet differentiable-quantization leg')
    ap.add_argument('--no-dp-configs', action='store_true', help='skip the ImageNet-shaped and seq2seq steps/sec legs')
    ap.add_argument('--precondition-s', type=float, default=0.4, help='seconds of untimed back-to-back launches before warm-up')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 PMC passes that measure the HBM traffic of the headline kernel')
    ap.add_argument('--deadline-s', type=float, default=1500.0,
                    help='after this many seconds rank 0 prints the line with what has been measured so far and the run ends')
    ap.add_argument('--skip-legs', default='', help='comma-separated steps/sec legs to leave out (cifar_student, diffquant_wrn, imagenet_resnet18k_dp, nmt_lstm_dp)')
    ap.add_argument('--quick', action='store_true', help='short steps/sec legs (a few steps, two repetitions): for exercising the flow, not for numbers')
    ap.add_argument('--pmc-slice', action='store_true', help=argparse.SUPPRESS)       # child mode of measure_pmc_traffic()
    ap.add_argument('--pmc-slice-launches', type=int, default=12, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_slice:
        pmc_slice(args.pmc_slice_launches)
        return

    from harness import launch, legs
    legs.rccl_env_defaults()                 # (before the first HIP call: the runtime reads HSA_ENABLE_IPC_MODE_LEGACY when it starts)
    if args.gpus > 1 and not launch.under_launcher():
        # `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, RCCL over xGMI
        if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
            raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible'
                             % (args.gpus, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        sys.stdout.flush()
        raise SystemExit(launch.run_ranks(os.path.abspath(__file__), args.gpus, sys.argv[1:]))

    # Everything that is not THE line goes to stderr: RCCL prints a version banner on the C-level stdout when its first
    # communicator comes up (also at N = 1 now that a one-rank group is always created), and the contract is one JSON
    # line on stdout.  File descriptor 1 points at stderr until the line is printed.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d\n'
                         % (args.gpus, world, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # QD_BENCH_BACKEND=gloo + QD_BENCH_ONE_GPU=1: the WHOLE multi-rank flow (per-leg agreement, the data-parallel report with
    # rank 0 alone, the closing barriers) on a one-GPU box -- every rank on device 0, collectives through gloo instead of RCCL
    # (tests/test_hip_bench_ranks.py).  The driver never sets these.
    backend = os.environ.get('QD_BENCH_BACKEND', 'nccl')
    if os.environ.get('QD_BENCH_ONE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    distributed = world > 1                  # the timed region's barriers: only where there is somebody to wait for
    rccl_error = None
    if launch.under_launcher():
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=legs.data_timeout())    # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend, timeout=legs.data_timeout())
        if world == 1:
            os.environ['QD_FORCE_DIST'] = '1'    # one rank under the launcher: the collectives are still issued (as in the branch below)
    else:
        # single process: still a (one-rank) RCCL group, and QD_FORCE_DIST=1 makes the harness issue its
        # collectives in it, so that the all-reduce path of the steps/sec legs is executed and timed on this box
        try:
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % launch.free_port(), rank=0, world_size=1,
                                    device_id=dev, timeout=legs.data_timeout())
            os.environ['QD_FORCE_DIST'] = '1'
        except Exception as e:                                 # noqa: BLE001 -- keep the headline measurement
            rccl_error = '%s: %s' % (type(e).__name__, e)
    n_gpus = world
    rccl_world_size = dist.get_world_size() if dist.is_initialized() else None
    runner = legs.LegRunner()                # the per-leg agreement runs over its own gloo group (harness/legs.py)

    # the line as far as it has been measured: what the deadline prints if the run does not finish
    line = {'metric': 'quantize_dequantize_GBps_64M_fp32_4bit', 'value': None, 'unit': 'GB/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic'}

    def emit(extra=None):
        d = dict(line)
        d.update(extra or {})
        os.write(saved_stdout_fd, (json.dumps(d) + '\n').encode())

    def expired():
        if rank == 0:
            emit({'error': 'deadline of %.0f s reached: the line holds what had been measured by then' % args.deadline_s,
                  'legs_failed': runner.history})
    deadline = legs.Deadline(args.deadline_s, expired)

    # Pre-flight: ONE small all-reduce through the data-path communicator before anything depends on it.  If RCCL cannot move
    # bytes on this node (no IPC path between the GPUs, a dead link) every rank learns it here, inside the group's timeout:
    # the headline then runs per GPU without the barriers of the timed region, the collective-bearing legs are skipped, and
    # rank 0's line says so -- instead of N ranks hanging in their first barrier.
    if distributed:
        pre_err = None
        try:
            if os.environ.get('QD_BENCH_TEST_PREFLIGHT_FAIL') == str(rank):      # test hook (tests/test_hip_bench_ranks.py)
                raise RuntimeError('pre-flight failure requested by the test')
            t = torch.ones(1, device=dev)
            dist.all_reduce(t)
            torch.cuda.synchronize()
            if int(t[0]) != world:
                pre_err = 'all-reduce of ones over %d ranks returned %r' % (world, float(t[0]))
        except Exception as e:                                    # noqa: BLE001
            pre_err = '%s: %s' % (type(e).__name__, e)
        failed = runner.agree(pre_err is None)
        if failed:
            rccl_error = 'pre-flight all-reduce failed on rank(s) %s%s' % (failed, ': ' + pre_err if pre_err else '')
            runner.broken = rccl_error
            distributed = False              # no collective in the timed region: every rank measures its own GPU
            line['error'] = rccl_error + '; value is rank 0 alone x n_gpus (NOT a max over ranks), the steps/sec legs were skipped'

    import quantization
    from quantized_distillation_amd import _lib
    _lib.load()

    gen = torch.Generator().manual_seed(1000 * rank)
    x_host = torch.randn(N_ELEM, generator=gen)
    xs = [x_host.to(dev)]
    for i in range(1, N_ROTATE):
        xs.append(torch.randn(N_ELEM, generator=gen).to(dev))
    live = [None] * N_ROTATE

    def step(i):
        q, _sf = quantization.uniformQuantization(xs[i % N_ROTATE], LEVELS, bucket_size=BUCKET)
        live[i % N_ROTATE] = q            # keep the last outputs alive: rotating output buffers

    # Preconditioning (setup, untimed): from an idle GPU the first ~25 ms of back-to-back
    # launches run 10-30 % slower while the power/clock management settles
    # (tools/sustain_probe.py, docs/history/profiles/r01_sustain_probe.txt); bring the chip to its steady
    # state before the official warm-up so that short --warmup/--steps runs measure steady state.
    t_pre = time.time()
    n_pre = 0
    while time.time() - t_pre < args.precondition_s:
        for i in range(50):
            step(i)
        n_pre += 50
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    event_ms_total = ev0.elapsed_time(ev1)

    if distributed:
        t = torch.tensor([elapsed, event_ms_total], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, event_ms_total = float(t[0]), float(t[1])

    # the same measurement over >= 200 launches whatever --steps says (the driver's --steps 20 region is 1.8 ms)
    ext_n = max(200, args.steps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(ext_n):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    ext_us = e0.elapsed_time(e1) * 1e3 / ext_n

    # reference point on this very GPU: torch's own device-to-device copy of the same 256 MiB tensors
    ya = torch.empty_like(xs[0])
    for _ in range(3):
        ya.copy_(xs[1])
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    c0.record()
    for i in range(20):
        live[i % N_ROTATE].copy_(xs[i % N_ROTATE])
    c1.record()
    torch.cuda.synchronize()
    copy_gbps = ALGO_BYTES_PER_ELEM * N_ELEM * 20 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del ya

    # What the call would cost a caller that hands over HOST buffers (the boundary takes device tensors: this is a note, never
    # `value`): pinned host -> device, quantize, device -> pinned host, 3 repetitions on one stream.
    pcie_gbps = None
    if rank == 0 and n_gpus == 1:
        try:
            hx, hq = x_host.pin_memory(), torch.empty(N_ELEM).pin_memory()
            xd = torch.empty_like(xs[0])
            ts = []
            for _ in range(4):
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                xd.copy_(hx, non_blocking=True)
                q_, _sf = quantization.uniformQuantization(xd, LEVELS, bucket_size=BUCKET)
                hq.copy_(q_, non_blocking=True)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t_a)
            pcie_gbps = ALGO_BYTES_PER_ELEM * N_ELEM / min(ts[1:]) / 1e9
            del hx, hq, xd, q_
        except Exception as e:                                    # noqa: BLE001 -- a note, not the measurement
            sys.stderr.write('pcie-inclusive note skipped: %s\n' % e)

    bytes_per_launch = ALGO_BYTES_PER_ELEM * N_ELEM
    kernel_us = event_ms_total * 1e3 / args.steps
    achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
    committed = load_pmc_traffic()
    roofline = {
        'bound': 'hbm', 'kernel': 'k_bucket_vec<MODE_QDQ,16,4>',
        'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBPS, 4),
        'traffic': int(round(committed)) if committed else None,
        'traffic_source': 'profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on a '
                          'builder box (committed file, NOT measured in this run)',
        'traffic_committed': int(round(committed)) if committed else None,
        'extended': {'launches': ext_n, 'avg_launch_us': round(ext_us, 3),
                     'frac': round(bytes_per_launch / (ext_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)},
        'extended_avg_launch_us': round(ext_us, 3),
        'avg_launch_us': round(kernel_us, 3), 'algorithmic_bytes_per_launch': bytes_per_launch,
        'timing': 'HIP events on the launch stream around the %d timed launches (includes inter-launch gaps)' % args.steps,
        'pcie_inclusive_GBps_note': round(pcie_gbps, 1) if pcie_gbps else None,   # host buffer -> HBM -> quantize -> host buffer (never `value`)
        'torch_d2d_copy_GBps': round(copy_gbps, 1),      # torch's own copy of the same bytes, same box (a hand-written NT copy reaches 6.3-6.5 TB/s: docs/history/profiles/r01_kbench.txt)
    }
    line.update({
        'value': round(bytes_per_launch * args.steps * n_gpus / elapsed / 1e9, 2),
        'ms_per_step': round(elapsed * 1e3 / args.steps, 5),
        'config': {
            'workload': 'uniformQuantization(x, s=16, bucket_size=256), x = randn(64Mi) fp32 per GPU (BASELINE configs[1] hot path, '
                        'headline size)',
            'call': 'quantization.uniformQuantization(x, s=16, type_of_scaling="linear", bucket_size=256), deterministic rounding, '
                    'through the public API (result allocation + one launch through the C ABI)',
            'n_elements_per_gpu': N_ELEM, 'levels': LEVELS, 'bucket_size': BUCKET,
            'algorithmic_bytes_per_element': ALGO_BYTES_PER_ELEM, 'rotating_buffers': N_ROTATE, 'precondition_launches': n_pre,
            'parallelism': 'independent tensors per rank (no data-path collective)' if n_gpus > 1 else 'single GPU',
        },
        'cpu_baseline': None, 'distill': None,
        'parity_bit_exact_vs_oracle': None, 'parity_bit_exact_vs_reference': None,
        'rccl_world_size': rccl_world_size, 'rccl_error': rccl_error, 'collective_backend': backend,
        'device': torch.cuda.get_device_name(dev),
        'roofline': roofline,                             # last: the driver keeps the tail of the line
    })

    # HBM traffic of the headline kernel, measured now (rank 0, N=1; after the timed regions, in a child process)
    if rank == 0 and n_gpus == 1 and not args.no_pmc:
        try:
            traffic_measured = measure_pmc_traffic()
        except Exception as e:                                    # noqa: BLE001
            traffic_measured = {'error': '%s: %s' % (type(e).__name__, e)}
        roofline['traffic_measured'] = traffic_measured
        if traffic_measured.get('bytes_per_launch'):
            roofline['traffic'] = int(traffic_measured['bytes_per_launch'])
            roofline['traffic_source'] = 'measured in this run (traffic_measured)'
            roofline['traffic_over_algorithmic'] = traffic_measured['over_algorithmic']
            for label, rec in (traffic_measured.get('other_kernels') or {}).items():        # scalars: what the driver's record keeps
                roofline['traffic_over_algorithmic ' + label] = rec['over_algorithmic']
        # ... and the kernel's duration as rocprofv3 sees it, next to the HIP-event figure above
        try:
            dur = measure_rocprof_duration()
        except Exception as e:                                    # noqa: BLE001
            dur = {'error': '%s: %s' % (type(e).__name__, e)}
        roofline['rocprof_kernel'] = dur
        if dur.get('avg_us'):
            roofline['rocprof_kernel_avg_us'] = dur['avg_us']
            roofline['rocprof_kernel_launches'] = dur['launches']
            roofline['rocprof_frac'] = round(bytes_per_launch / (dur['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)

    # cpu_baseline leg (rank 0, N=1 only): the oracle is timed on the host cores and, in the same leg,
    # used as the checker of the GPU result computed above (bit-exact comparison)
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        q, sf = quantization.uniformQuantization(xs[0], LEVELS, bucket_size=BUCKET)
        try:
            cpu = cpu_baseline(x_host, q.cpu().numpy(), sf.alpha.cpu().numpy().reshape(-1))
            line['parity_bit_exact_vs_oracle'] = cpu.pop('gpu_result_bit_exact')
            line['parity_bit_exact_vs_reference'] = cpu.pop('gpu_result_bit_exact_vs_reference', None)
        except Exception as e:                                    # noqa: BLE001  (keep the headline; say what failed)
            import traceback
            sys.stderr.write(traceback.format_exc())
            cpu = {'error': '%s: %s' % (type(e).__name__, e)}
        line['cpu_baseline'] = cpu
        del q, sf
    del live[:], xs[:]
    torch.cuda.empty_cache()

    # Per-kernel roofline rows (every rank measures its own GPU; rank 0's rows are reported): no collective inside.
    if not args.no_kernels:
        from harness import kernel_bench
        rows = runner.run('kernels', kernel_bench.run, dev, collective=False)
        roofline['kernels'] = rows
        roofline['kernels_method'] = ('HIP events, median of %d repetitions of %d launches (12 for the 1 Gi-symbol histograms) after >= 100 ms '
                                      'of preconditioning, >= 3 rotating buffer sets' % (kernel_bench.REPS, kernel_bench.ITERS))
        torch.cuda.empty_cache()

    # The steps/sec legs are reported alongside the headline.  A failure in one of them (say, an out-of-memory condition
    # on one rank) is recorded in the JSON instead of losing the headline measurement above, and cannot leave the other
    # ranks inside a collective for longer than the group's timeout: runner.run() wraps the leg BODY and ends every leg
    # with an agreement over a gloo group (harness/legs.py).
    if not args.no_distill:
        distill = {}
        line['distill'] = distill
        quick = dict(steps=3, warmup=2, reps=2) if args.quick else {}
        skip = set(x for x in args.skip_legs.split(',') if x)
        if 'cifar_student' not in skip:
            distill['cifar_student'] = runner.run('cifar_student', distill_steps_per_sec, dev, rank, n_gpus, distributed, runner.barrier,
                                                  **(dict(steps=20, warmup=5, repetitions=2) if args.quick else {}))
            torch.cuda.empty_cache()
        if not args.no_diffquant and 'diffquant_wrn' not in skip:
            distill['diffquant_wrn'] = runner.run('diffquant_wrn', diffquant_steps_per_sec, dev, rank, n_gpus, distributed, runner.barrier, **quick)
            torch.cuda.empty_cache()
        # configs[3] is quoted on 8 GPUs and configs[4] on 4; both fit one GPU, so they are timed at every N
        # (weak scaling: per-GPU batch fixed) and the driver's --gpus 8 / --gpus 4 runs give BASELINE's placements
        if not args.no_dp_configs and 'imagenet_resnet18k_dp' not in skip:
            distill['imagenet_resnet18k_dp'] = runner.run('imagenet_resnet18k_dp', dp_config_steps_per_sec, 'imagenet', dev, rank, n_gpus,
                                                          distributed, runner.barrier, **quick)
            torch.cuda.empty_cache()
        if not args.no_dp_configs and 'nmt_lstm_dp' not in skip:
            distill['nmt_lstm_dp'] = runner.run('nmt_lstm_dp', dp_config_steps_per_sec, 'nmt', dev, rank, n_gpus, distributed, runner.barrier, **quick)
        if runner.history:
            distill['legs_failed'] = [{'leg': n, 'ranks': r} for n, r in runner.history]

    # the scalars the driver's record keeps (it drops nested objects): steps/sec + data-parallel figures, then the kernel rows
    if line['distill']:
        d = line['distill']
        cs = d.get('cifar_student') or {}
        if 'multi' in cs:
            roofline['steps_cfg1'] = ' | '.join('%s %.1f (%.1f-%.1f)' % (m, cs[m]['steps_per_sec'], cs[m]['steps_per_sec_min'], cs[m]['steps_per_sec_max'])
                                                for m in ('multi_graph', 'multi', 'per_tensor') if 'steps_per_sec' in (cs.get(m) or {}))[:118]
            roofline['dp_cfg1'] = flat_dp(cs.get('dp'))
        else:
            roofline['steps_cfg1'] = flat_dp(cs)
        for key, tag in (('diffquant_wrn', 'dp_cfg2_wrn_diffquant'), ('imagenet_resnet18k_dp', 'dp_cfg3_imagenet'), ('nmt_lstm_dp', 'dp_cfg4_nmt')):
            if key in d:
                roofline[tag] = flat_dp(d[key])
    if isinstance(roofline.get('kernels'), list):
        from harness.kernel_bench import flat_row
        for i, r in enumerate(roofline['kernels']):
            roofline['k%02d' % (i + 1)] = flat_row(r)

    # RCCL writes a version banner to the C-level stdout, which is block-buffered when piped and would
    # otherwise be flushed at exit, i.e. after the JSON line: push it out now on every rank, so that the JSON
    # is the last thing on stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    runner.barrier()                                              # (gloo: works whatever the legs left behind)
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    deadline.cancel()
    if rank == 0:
        emit()
    os.dup2(saved_stdout_fd, 1)                                   # stdout is stdout again
    if dist.is_initialized():
        runner.barrier()
        if runner.broken is None:
            dist.destroy_process_group()
        else:
            os._exit(0)                                           # a communicator with unmatched collectives cannot be torn down cleanly


if __name__ == '__main__':
    main()
