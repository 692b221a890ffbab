"""The legs of bench.py: what is measured, one function per leg (bench.py itself is orchestration only).

    headline / roofline       bench.py (the timed region of the contract)
    rocprof + PMC children    measure_rocprof_duration(), measure_pmc_traffic()  (child processes under rocprofv3)
    cpu_baseline              cpu_baseline(): the REFERENCE's own quantizer on the host cores, same workload
    cifar_student             distill_steps_per_sec(): BASELINE configs[1], eager, with the data-parallel report
    cifar_graph               distill_graph_steps_per_sec(): the same step replayed from hipGraphs
    diffquant_wrn             diffquant_steps_per_sec(): configs[2]
    imagenet / nmt            dp_config_steps_per_sec(): configs[3], configs[4]
    cpu_distill               cpu_distill_baseline(): configs[0], the CPU reference path

Only cpu_baseline / cpu_distill import anything under oracle/ (as the checker and the timed CPU baseline).
"""
import json
import os
import statistics
import sys
import time

import torch

from .dpbench import XGMI_PEAK_GBPS, dp_report, event_ms, flat_dp, timed_steps  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')

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


def _time_runs(fn, min_runs, budget_s, warm=True):
    """[Warm-up +] >= min_runs timed runs (more while the time budget lasts, at most 10)."""
    if warm:
        fn()
    ts = []
    t_end = time.time() + budget_s
    while len(ts) < min_runs or (len(ts) < 10 and time.time() < t_end):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts



def cpu_baseline(x_host, q_gpu, alpha_gpu, budget_s=2.0, with_ports=True):
    """The reference's own quantizer on the host cores of this box, same workload (bounded sample: ~10 s of CPU work);
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
        # the baseline; os.cpu_count() threads -- what the survey prescribes -- is always among them (bounded: a count
        # whose single run takes about a second or more gets ONE timed run after its warm-up, not five).
        counts = sorted({ncpu, min(ncpu, 64), min(ncpu, 32)}, reverse=True)
        per_threads, best = {}, None
        for th in counts:
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            refq.uniformQuantization(x_host, LEVELS, bucket_size=BUCKET)            # warm-up, and a first idea of the cost
            first = time.perf_counter() - t0
            ts = _time_runs(lambda: refq.uniformQuantization(x_host, LEVELS, bucket_size=BUCKET), 1 if first > 0.8 else 5,
                            budget_s, warm=False)
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
            'sample_short': "%d runs of the full 64Mi-element call by the reference's own uniformQuantization (oracle/_ref), torch CPU ops, "
                            'best of %s threads; min %.4f s' % (best[3], '/'.join(str(c) for c in counts), best[1]),
            'threads_tried': per_threads,
            'reference_sources_sha256': (ref_stage.manifest() or {}).get('files'),
        })
        bit_exact = bool(np.array_equal(q_gpu, q_ref.numpy()) and
                         np.array_equal(alpha_gpu, sf_ref.alpha.numpy().reshape(-1)))
        out['gpu_result_bit_exact_vs_reference'] = bit_exact
        # next to the baseline, NOT a baseline: what the PRODUCT does with the same CPU tensor (libqd_host.so, the library CPU
        # tensors are computed by -- the same API call on a tensor that lives on the host), and that its result is the reference's
        try:
            import quantization
            from quantized_distillation_amd import _lib
            th = min(ncpu, 64)
            _lib.host().qd_host_set_threads(th)
            tp = _time_runs(lambda: quantization.uniformQuantization(x_host, LEVELS, bucket_size=BUCKET), 3, 1.0)
            q_host, sf_host = quantization.uniformQuantization(x_host, LEVELS, bucket_size=BUCKET)
            out['product_on_cpu_tensors'] = {
                'value': round(ALGO_BYTES_PER_ELEM * n / min(tp) / 1e9, 3), 'unit': 'GB/s', 'threads': th,
                'bit_exact_vs_reference': bool(torch.equal(q_host, q_ref) and torch.equal(sf_host.alpha, sf_ref.alpha)),
                'sample': '%d runs, min %.4f s; quantization.uniformQuantization on the CPU tensor -> libqd_host.so (csrc/host/qd_host.cpp, OpenMP)'
                          % (len(tp), min(tp))}
            del q_host, sf_host
        except Exception as e:                                    # noqa: BLE001 -- a side figure
            out['product_on_cpu_tensors'] = {'error': '%s: %s' % (type(e).__name__, e)}
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
    if with_ports or 'value' not in out:
        ts = _time_runs(lambda: oracle_c.uniform_quantize(xn, LEVELS, BUCKET, want_idx=False, want_lev=False), 3, 1.0, warm=False)
        out['c_port'] = {'value': round(ALGO_BYTES_PER_ELEM * n / min(ts) / 1e9, 3), 'unit': 'GB/s', 'threads': cores,
                         'sample': '%d runs, min %.4f s, median %.4f s; oracle/qd_oracle.c, OpenMP over buckets'
                                   % (len(ts), min(ts), float(np.median(ts)))}
        if 'value' not in out:
            out.update({'value': out['c_port']['value'], 'cores': cores, 'kind': 'port', 'sample': out['c_port']['sample']})
    if with_ports:
        torch.set_num_threads(min(ncpu, 64))
        tt = _time_runs(lambda: uniform_quantize_torch_ops(x_host, LEVELS, BUCKET), 3, 1.0)
        out['torch_ops_port'] = {
            'value': round(ALGO_BYTES_PER_ELEM * n / min(tt) / 1e9, 3), 'unit': 'GB/s', 'threads': torch.get_num_threads(),
            'sample': '%d runs, min %.4f s, median %.4f s; same sequence of torch CPU ops as '
                      'quantization/quant_functions.py:155-194 (oracle/torch_port.py)' % (len(tt), min(tt), float(np.median(tt))),
        }
    return out


def cpu_distill_baseline(steps=60, warmup=3, batch=50):
    """BASELINE configs[0]: the CIFAR10 ConvolForwardNet student step on the CPU in the reference's loop shape (quantize
    every parameter, fwd/bwd with the KD loss, restore, SGD) -- with the reference's own quantizer (staged bytecode; its
    torch-op port when nothing is staged): THE BASELINE -- and, beside it, the same loop with this package's quantizer on
    the same CPU tensors (libqd_host.so): the product on configs[0].  A bounded sample (`steps` of the 200 steps of
    configs[0]'s "1 epoch synthetic" = 10000 images / batch 50, BASELINE.md 4.4, split between the two; --cpu-distill-steps
    400 runs the whole epoch for each): steps/sec does not depend on how many are timed."""
    from . import models
    from oracle import ref_stage
    from oracle.torch_port import uniform_quantize_torch_ops
    refq = ref_stage.load()
    if refq is not None:
        def reference_quantizer(t):
            return refq.uniformQuantization(t, 16, bucket_size=256)[0]
    else:
        def reference_quantizer(t):
            return uniform_quantize_torch_ops(t, 16, 256)[0]

    def product_quantizer(t):
        import quantization
        return quantization.uniformQuantization(t, 16, bucket_size=256)[0]
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    try:
        from quantized_distillation_amd import _lib
        _lib.host().qd_host_set_threads(threads)
    except Exception:                                             # noqa: BLE001 -- the product leg below reports it
        pass
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(batch, 3, 32, 32, generator=g), torch.randint(0, 10, (batch,), generator=g)

    def run(quantize_one, n_steps):
        torch.manual_seed(0)
        st, te = models.student().train(), models.teacher().eval()
        opt = torch.optim.SGD(st.parameters(), lr=1e-3, momentum=0.9, nesterov=True, weight_decay=2.2e-4)
        t_quant = [0.0]

        def one():
            a = time.perf_counter()
            saved = [p.data for p in st.parameters()]
            for p in st.parameters():
                p.data = quantize_one(p.data)
            t_quant[0] += time.perf_counter() - a
            opt.zero_grad()
            with torch.no_grad():
                t_out = te(x)
            models.kd_loss(st(x), t_out, y).backward()
            for p, m in zip(st.parameters(), saved):
                p.data = m
            opt.step()

        for _ in range(warmup):
            one()
        t_quant[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(n_steps):
            one()
        dt = time.perf_counter() - t0
        final = torch.cat([p.detach().reshape(-1) for p in st.parameters()])
        return {'steps_per_sec': round(n_steps / dt, 3), 'ms_per_step': round(dt / n_steps * 1e3, 2),
                'quantize_ms_per_step': round(t_quant[0] / n_steps * 1e3, 3), 'steps': n_steps}, final

    half = max(1, steps // 2)
    base, w_ref = run(reference_quantizer, half)
    out = dict(base)
    out.update({'threads': threads,
                'quantizer': 'reference (oracle/_ref bytecode)' if refq is not None else 'torch-op port (oracle/torch_port.py)',
                'sample': '%d steps (of the 200 of one synthetic epoch: 10000 images) after %d warm-up steps, batch %d, synthetic '
                          'CIFAR10-shaped data; student+teacher fwd, KD loss, bwd, SGD on the host with the reference quantizer in the '
                          'loop (configs[0])' % (half, warmup, batch)})
    try:
        prod, w_prod = run(product_quantizer, half)
        prod['quantizer'] = 'this package on the same CPU tensors (libqd_host.so)'
        prod['weights_after_training_bit_identical_to_the_reference_run'] = bool(torch.equal(w_ref, w_prod))
        out['product_on_cpu_tensors'] = prod
    except Exception as e:                                        # noqa: BLE001
        out['product_on_cpu_tensors'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


CIFAR_DESC = ('CIFAR10-shaped synthetic randn(B,3,32,32), ConvolForwardNet student (22 tensors, 1.00 M params) distilled from the '
              '5.3 M teacher, KD loss T=2, SGD nesterov, 4-bit uniform, bucket 256, STE')


def _rep_stats(reps, steps, per_gpu_batch, n_gpus):
    dt = statistics.median(reps)
    sps = sorted(steps / r for r in reps)
    return {'steps_per_sec': round(steps / dt, 2), 'ms_per_step': round(dt / steps * 1e3, 4),
            'samples_per_sec': round(steps * per_gpu_batch * n_gpus / dt, 1), 'statistic': 'median of %d repetitions' % len(reps),
            'steps_per_sec_min': round(sps[0], 1), 'steps_per_sec_max': round(sps[-1], 1),
            'steps_per_sec_repetitions': [round(steps / r, 1) for r in reps]}


def _cifar_trainer(dev, mode, warmup, batches):
    from . import models
    from .distill import DistillTrainer
    torch.manual_seed(0)                                   # identical replicas on every rank
    tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode=mode)
    for i in range(warmup):
        tr.step(*batches[i % 4])
    return tr


def distill_steps_per_sec(dev, rank, n_gpus, distributed, ctl_barrier, steps=100, warmup=20, per_gpu_batch=50, repetitions=5):
    """Second half of BASELINE.json's metric: distilled-training steps/sec on synthetic
    CIFAR10-shaped data (configs[1]: ConvolForwardNet student, 4-bit uniform quantization, bucket
    256, pure STE), data parallel over the ranks with one RCCL all-reduce of the flat gradient
    per step.  Weak scaling: per-GPU batch fixed at 50.  Eager launches only (the hipGraph replay of
    the same step is its own, optional leg: distill_graph_steps_per_sec)."""
    from .distill import synthetic_batch
    out = {'config': CIFAR_DESC, 'per_gpu_batch': per_gpu_batch, 'global_batch': per_gpu_batch * n_gpus, 'steps': steps, 'warmup': warmup}
    modes = ('multi', 'per_tensor')
    batches = [synthetic_batch(per_gpu_batch, dev, seed=1000 * rank + i) for i in range(4)]
    trainers = {mode: _cifar_trainer(dev, mode, warmup, batches) for mode in modes}
    # The 2 ms step is ~150 small launches (MIOpen's small-shape convolutions, batch-norm, the optimizer) and its time
    # moves from repetition to repetition on one box (docs/history/profiles/r03_distill_spread.txt).  So: REPS repetitions of `steps`
    # steps per mode, INTERLEAVED (multi, per_tensor, multi, ...) so that drift hits both alike; the MEDIAN is reported,
    # every repetition is listed, and the two modes are only called different when their ranges do not overlap.
    reps = {m: [] for m in modes}
    for _rep in range(repetitions):
        for mode in modes:
            tr = trainers[mode]
            job, _own = timed_steps(lambda i, tr=tr: tr.step(*batches[i % 4]), steps, 1, dev, distributed)
            reps[mode].append(job[0])
    for mode in modes:
        tr = trainers[mode]
        out[mode] = _rep_stats(reps[mode], steps, per_gpu_batch, n_gpus)
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
    # the data-parallel figures of this config, on the eager multi-tensor trainer
    tr = trainers['multi']

    def set_exchange(on, tr=tr):
        tr.sync.active = on and tr.sync.world_active
    out['dp'] = dp_report(lambda i: tr.step(*batches[i % 4]), steps, 3, dev, n_gpus, distributed, per_gpu_batch,
                          tr.flat_grad.numel() * 4, set_exchange, tr.sync.sync if tr.sync.active else None, ctl_barrier, rank)
    out['dp']['trainer'] = 'multi'
    trainers.clear()
    out['note'] = ("'multi' = one multi-tensor quantize launch per step on persistent shadows (K9); 'per_tensor' = the "
                   "reference's loop shape (22 uniformQuantization calls + restore)")
    return out


def distill_graph_steps_per_sec(dev, rank, n_gpus, distributed, steps=100, warmup=20, per_gpu_batch=50, repetitions=5):
    """The configs[1] step replayed from hipGraphs (quantize + forward + loss + backward in one graph, the optimizer in a
    second, the RCCL all-reduce eager between the two), interleaved with the eager multi-tensor step: the ~150 launches of
    a step stop depending on the host, which is where the repetition-to-repetition spread of the eager legs comes from.
    Optional leg: it runs after the line is safe, and by default only at N = 1 (DistillTrainer.capture is thread-local,
    drains the collectives first and restores the stream on failure -- tests/test_hip_capture_watchdog.py -- but no
    multi-rank RCCL box has run it yet).  Every rank must take the same branch: the capture verdict is agreed on first."""
    import torch.distributed as dist
    from .distill import synthetic_batch
    batches = [synthetic_batch(per_gpu_batch, dev, seed=1000 * rank + i) for i in range(4)]
    out = {'config': CIFAR_DESC, 'per_gpu_batch': per_gpu_batch, 'steps': steps}
    te = _cifar_trainer(dev, 'multi', warmup, batches)
    tg = _cifar_trainer(dev, 'multi', warmup, batches)
    err = None
    try:
        tg.capture(*batches[0])                             # local: no collective inside
    except Exception as e:                                  # noqa: BLE001 -- reported; tg stays an eager trainer and is dropped
        err = '%s: %s' % (type(e).__name__, e)
    if distributed:
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            err = err or 'graph capture failed on another rank'
    if err is not None:
        return {'error': err}
    for i in range(warmup):
        tg.step(*batches[i % 4])
    reps = {'multi_graph': [], 'multi': []}
    for _rep in range(repetitions):
        for name, tr in (('multi_graph', tg), ('multi', te)):
            job, _own = timed_steps(lambda i, tr=tr: tr.step(*batches[i % 4]), steps, 1, dev, distributed)
            reps[name].append(job[0])
    for name in reps:
        out[name] = _rep_stats(reps[name], steps, per_gpu_batch, n_gpus)
    lo_g, hi_g = out['multi_graph']['steps_per_sec_min'], out['multi_graph']['steps_per_sec_max']
    lo_m, hi_m = out['multi']['steps_per_sec_min'], out['multi']['steps_per_sec_max']
    out['multi_graph_vs_multi'] = ('graph replay faster in every repetition' if lo_g > hi_m else
                                   'eager faster in every repetition' if lo_m > hi_g else
                                   'indistinguishable: the repetition ranges overlap')
    out['note'] = ("'multi_graph' = the 'multi' step replayed from two hipGraphs (DistillTrainer.capture; "
                   "tests/test_hip_distill.py::test_graph_replay_matches_eager): the GPU work is the same MIOpen small-shape kernels, "
                   "but ~150 launches per step no longer wait for the host, so the repetitions stop spreading")
    return out


def pcie_inclusive_note(x_host, dev):
    """What the headline call costs a caller that hands over HOST buffers (the boundary takes device tensors: this is a note,
    never `value`): pinned host -> device, quantize, device -> pinned host, 3 repetitions on one stream.  GB/s."""
    import quantization
    hx, hq = x_host.pin_memory(), torch.empty(x_host.numel()).pin_memory()
    xd = torch.empty(x_host.numel(), device=dev)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        xd.copy_(hx, non_blocking=True)
        q_, _sf = quantization.uniformQuantization(xd, LEVELS, bucket_size=BUCKET)
        hq.copy_(q_, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t_a)
    return round(ALGO_BYTES_PER_ELEM * x_host.numel() / min(ts[1:]) / 1e9, 1)


def dp_config_steps_per_sec(kind, dev, rank, n_gpus, distributed, ctl_barrier, steps=10, warmup=3, reps=3):
    """BASELINE configs[3] (kind='imagenet': ImageNet-shaped synthetic, resnet_kfilters
    resnet18(k=1.5) student distilled from a ResNet-34-shaped teacher, 4-bit bucketed, first/last
    tensors not quantized, DP over 8 GPUs) and configs[4] (kind='nmt': 2-layer LSTM seq2seq,
    multi30k-shaped synthetic tokens, 4-bit quantized distillation, DP over 4 GPUs).  Data
    parallel with the flat-gradient RCCL all-reduce, cut in 4 pieces overlapped with backward."""
    from . import models
    from .distill import (DistillTrainer, seq2seq_kd_loss_fn, synthetic_batch, synthetic_token_batch)
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
    t_w = time.perf_counter()
    for i in range(warmup):
        tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    warmup_s = time.perf_counter() - t_w

    def set_exchange(on):
        tr.sync.active = on and tr.sync.world_active
    out = {'config': desc, 'warmup_s (first use: MIOpen searches its plans for these shapes)': round(warmup_s, 1)}
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
    from . import models
    from .diffquant import DiffQuantTrainer
    from .distill import synthetic_batch
    torch.manual_seed(0)
    t0 = time.perf_counter()
    tr = DiffQuantTrainer(models.WideResNet(16, 22), dev, num_points=4, bucket_size=256, lr=1e-5, mode='multi')
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    x, y = synthetic_batch(batch, dev, seed=11 + 1000 * rank)
    t_w = time.perf_counter()
    for _ in range(warmup):
        tr.step(x, y)
    torch.cuda.synchronize()
    warmup_s = time.perf_counter() - t_w
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
    out['warmup_s (first use: MIOpen searches its plans for these shapes)'] = round(warmup_s, 1)
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
               sys.executable, BENCH, '--pmc-slice', '--pmc-slice-launches', str(launches)]
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
                   sys.executable, BENCH, '--pmc-slice']
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
