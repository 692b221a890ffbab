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
ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      achieved algorithmic GB/s of the dominant kernel (8 B/element: 4 read + 4 written,
                SURVEY.md 8d) from HIP-event timing of the timed region, against the 8 TB/s peak
  cpu_baseline  the same algorithm on the host cores (C port of the reference, OpenMP) on a
                bounded sample, and the op-for-op torch CPU port of the reference next to it.
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


def cpu_baseline(x_host, q_gpu, alpha_gpu):
    """The reference's algorithm on the host cores, same workload, bounded sample; the oracle's
    output doubles as the checker of the GPU result for the same tensor (q_gpu, alpha_gpu)."""
    import numpy as np
    from oracle import oracle_c
    from oracle.torch_port import uniform_quantize_torch_ops
    oracle_c.build()
    cores = oracle_c.max_threads()
    xn = x_host.numpy()
    n = xn.size
    ref = oracle_c.uniform_quantize(xn, LEVELS, BUCKET, want_idx=False, want_lev=False)       # warm-up + checker
    bit_exact = bool(np.array_equal(q_gpu, ref['q']) and np.array_equal(alpha_gpu, ref['alpha']))
    del ref
    times = []
    t_end = time.time() + 12.0
    while len(times) < 10 and (time.time() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        oracle_c.uniform_quantize(xn, LEVELS, BUCKET, want_idx=False, want_lev=False)
        times.append(time.perf_counter() - t0)
    best, med = min(times), float(np.median(times))
    out = {
        'value': round(ALGO_BYTES_PER_ELEM * n / best / 1e9, 3), 'unit': 'GB/s', 'cores': cores, 'kind': 'port',
        'sample': '%d runs of the full workload (N=%d fp32, s=%d, bucket=%d); C port of the reference '
                  'algorithm (oracle/qd_oracle.c, OpenMP over buckets); min %.4f s, median %.4f s'
                  % (len(times), n, LEVELS, BUCKET, best, med),
        'cpu_model': cpu_model(), 'os_cpu_count': os.cpu_count(), 'gpu_result_bit_exact': bit_exact,
        'port_vs_reference': 'on identical cores (authoring container, profiles/r01_reference_cpu_timing.json) the '
                             'reference itself runs this workload at 5.2 GB/s, torch_ops_port at 4.9 (0.95 x), this C '
                             'port at 8.5 (1.64 x); both ports are bit-identical to the reference output',
    }
    # the reference's own op chain (multi-threaded torch CPU ops), restated in oracle/torch_port.py
    torch.set_num_threads(min(os.cpu_count() or 1, 64))     # 256 SMT threads oversubscribe torch's elementwise ops
    uniform_quantize_torch_ops(x_host, LEVELS, BUCKET)
    tt = []
    t_end = time.time() + 12.0
    while len(tt) < 5 and (time.time() < t_end or len(tt) < 2):
        t0 = time.perf_counter()
        uniform_quantize_torch_ops(x_host, LEVELS, BUCKET)
        tt.append(time.perf_counter() - t0)
    out['torch_ops_port'] = {
        'value': round(ALGO_BYTES_PER_ELEM * n / min(tt) / 1e9, 3), 'unit': 'GB/s',
        'threads': torch.get_num_threads(),
        'sample': '%d runs, min %.4f s, median %.4f s; same sequence of torch CPU ops as '
                  'quantization/quant_functions.py:155-194' % (len(tt), min(tt), float(np.median(tt))),
    }
    out['distill'] = cpu_distill_baseline()
    return out


def cpu_distill_baseline(steps=12, warmup=2, batch=50):
    """BASELINE configs[0]: the CIFAR10 ConvolForwardNet student step on the CPU with the
    reference's quantizer (its torch-op chain, oracle/torch_port.py) in the reference's loop shape
    (quantize every parameter, fwd/bwd with the KD loss, restore, SGD) -- bounded sample."""
    from harness import models
    from oracle.torch_port import uniform_quantize_torch_ops
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
            p.data = uniform_quantize_torch_ops(p.data, 16, 256)[0]
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
            'sample': '%d steps, batch %d, synthetic CIFAR10-shaped data; student+teacher fwd, KD loss, bwd, SGD on the '
                      'host with the torch-op port of the reference quantizer (configs[0])' % (steps, batch)}


def distill_steps_per_sec(dev, rank, n_gpus, distributed, steps=100, warmup=20, per_gpu_batch=50):
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
    for mode in ('multi', 'per_tensor'):
        tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256,
                            mode='multi' if mode == 'multi_graph' else mode)
        batches = [synthetic_batch(per_gpu_batch, dev, seed=1000 * rank + i) for i in range(4)]
        if mode == 'multi_graph':
            try:
                tr.capture(*batches[0])
            except Exception as e:                         # noqa: BLE001 -- report, do not hide
                out[mode] = {'error': 'graph capture failed: %r' % (e,)}
                del tr
                continue
        for i in range(warmup):
            tr.step(*batches[i % 4])
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(*batches[i % 4])
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        # per-phase breakdown (each phase bracketed by synchronize; serialised, so the sum exceeds the step)
        phases = {}
        def timed(name, fn, reps=20):
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            phases[name] = round((time.perf_counter() - a) / reps * 1e3, 4)
        x, y = batches[0]
        timed('quantize_ms', tr.quantize)
        timed('fwd_bwd_ms', lambda: tr.forward_backward(x, y))
        timed('restore_ms', tr.restore)
        timed('allreduce_ms', tr.sync.sync)
        timed('optimizer_ms', tr.opt.step)
        out[mode] = {'steps_per_sec': round(steps / dt, 2), 'ms_per_step': round(dt / steps * 1e3, 4),
                     'samples_per_sec': round(steps * per_gpu_batch * n_gpus / dt, 1), 'phases': phases}
        del tr
    out['note'] = ("'multi' = one multi-tensor quantize launch per step on persistent shadows (K9); 'per_tensor' = the "
                   "reference's loop shape (22 uniformQuantization calls + restore).  hipGraph replay of the step "
                   "(DistillTrainer.capture) measured no gain: the step is bound by MIOpen's small-shape conv kernels, "
                   "not by launches (profiles/r01_distill_notes.txt)")
    return out


def dp_config_steps_per_sec(kind, dev, rank, n_gpus, distributed, steps=10, warmup=3):
    """BASELINE configs[3] (kind='imagenet': ImageNet-shaped synthetic, resnet_kfilters
    resnet18(k=1.5) student distilled from a ResNet-34-shaped teacher, 4-bit bucketed, first/last
    tensors not quantized, DP over 8 GPUs) and configs[4] (kind='nmt': 2-layer LSTM seq2seq,
    multi30k-shaped synthetic tokens, 4-bit quantized distillation, DP over 4 GPUs).  Data
    parallel with the flat-gradient RCCL all-reduce, cut in 4 pieces overlapped with backward."""
    import torch.distributed as dist
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
                            grad_chunks=4, overlap_allreduce=True)
        batches = [synthetic_token_batch(per_gpu, dev, seed=1000 * rank + i) for i in range(2)]
        desc = ('multi30k-shaped synthetic tokens (len 20..50, V_src 18000, V_tgt 10000), 2-layer LSTM 500/500 with input '
                'feeding + general attention (22 tensors, 28.8 M), teacher of the same shape, word-level KD 0.3 NLL + 0.7 KL; '
                'SGD lr 1.0, clip-norm 5; 4-bit uniform, bucket 256')
    for i in range(warmup):
        tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    phases = {}

    def timed(name, fn, reps=5):
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        phases[name] = round((time.perf_counter() - a) / reps * 1e3, 3)
    timed('quantize_ms', tr.quantize)
    timed('fwd_bwd_ms (+overlapped all-reduce launch)', lambda: (tr.forward_backward(*batches[0]), tr.sync.sync()))
    timed('optimizer_ms', tr.opt.step)
    out = {'config': desc, 'per_gpu_batch': per_gpu, 'global_batch': per_gpu * n_gpus, 'n_gpus': n_gpus,
           'steps_per_sec': round(steps / dt, 3), 'ms_per_step': round(dt / steps * 1e3, 2),
           'samples_per_sec': round(steps * per_gpu * n_gpus / dt, 1), 'steps': steps, 'phases': phases,
           'gradient_bytes_per_step': int(tr.flat_grad.numel() * 4)}
    del tr
    torch.cuda.empty_cache()
    return out


def diffquant_steps_per_sec(dev, steps=8, warmup=2, batch=100):
    """BASELINE configs[2]: CIFAR10 WideResNet-16-22 student (60 tensors, 82.7 M parameters), 2-bit
    (k = 4 points) non-uniform differentiable quantization, bucket 256, 1 GPU: steps/sec of the
    optimize_quantization_points loop with the per-step quantizer cost broken out."""
    from harness import models
    from harness.diffquant import DiffQuantTrainer
    from harness.distill import synthetic_batch
    torch.manual_seed(0)
    t0 = time.perf_counter()
    tr = DiffQuantTrainer(models.WideResNet(16, 22), dev, num_points=4, bucket_size=256, lr=1e-5, mode='multi')
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    x, y = synthetic_batch(batch, dev, seed=11)
    for _ in range(warmup):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    phases = {}

    def timed(name, fn, reps=5):
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        phases[name] = round((time.perf_counter() - a) / reps * 1e3, 3)
    timed('assign_all_tensors_ms (multi-tensor K5, 1 launch)', tr.quantize)
    timed('fwd_bwd_ms', lambda: tr.forward_backward(x, y))
    timed('point_gradients_ms (multi-tensor K6, 2 launches)', tr.point_gradients)
    nparams = sum(p.numel() for p in tr.params)
    return {'config': 'Wide_ResNet depth 16 widen 22 (60 tensors, %.1f M params), k=4 points (2-bit) per tensor, bucket 256, '
                      'percentile init, KD loss vs the unquantized model, SGD on the points; batch %d synthetic CIFAR10-shaped'
                      % (nparams / 1e6, batch),
            'steps_per_sec': round(steps / dt, 3), 'ms_per_step': round(dt / steps * 1e3, 2), 'steps': steps,
            'setup_s': round(setup_s, 2), 'phases': phases,
            'reference_cpu_quantizer_note': 'reference per-step quantizer cost on this model, CPU path: ~2.5 s per 16 Mi-element '
                                            'tensor (BASELINE.md section 3); here the 60-tensor assign + point-gradient pair is the '
                                            'two phase entries above'}


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if present."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get('k_bucket_vec_hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-distill', action='store_true', help='skip the distilled-training steps/sec leg')
    ap.add_argument('--no-diffquant', action='store_true', help='skip the WideResNet differentiable-quantization leg')
    ap.add_argument('--precondition-s', type=float, default=0.4, help='seconds of untimed back-to-back launches before warm-up')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # QD_FORCE_DIST=1 exercises the RCCL path (init, barrier, all-reduce) even with a single rank
    distributed = world > 1 or os.environ.get('QD_FORCE_DIST') == '1'
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)        # "nccl" is RCCL on ROCm
    n_gpus = world if distributed else 1

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
    # (tools/sustain_probe.py, profiles/r01_sustain_probe.txt); bring the chip to its steady
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
    event_ms = ev0.elapsed_time(ev1)

    if distributed:
        t = torch.tensor([elapsed, event_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, event_ms = float(t[0]), float(t[1])

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

    # cpu_baseline leg (rank 0, N=1 only): the oracle is timed on the host cores and, in the same leg,
    # used as the checker of the GPU result computed above (bit-exact comparison)
    parity = None
    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        q, sf = quantization.uniformQuantization(xs[0], LEVELS, bucket_size=BUCKET)
        try:
            cpu = cpu_baseline(x_host, q.cpu().numpy(), sf.alpha.cpu().numpy().reshape(-1))
            parity = cpu.pop('gpu_result_bit_exact')
        except Exception as e:                                    # noqa: BLE001  (keep the headline; say what failed)
            import traceback
            sys.stderr.write(traceback.format_exc())
            cpu = {'error': '%s: %s' % (type(e).__name__, e)}
        del q, sf

    # The steps/sec legs are reported alongside the headline; a failure in one of them (say, an out-of-memory
    # condition on a shared box) is recorded in the JSON instead of losing the headline measurement above.
    def leg(fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as e:                                    # noqa: BLE001
            import traceback
            sys.stderr.write(traceback.format_exc())
            return {'error': '%s: %s' % (type(e).__name__, e)}

    distill = None
    if not args.no_distill:
        del live[:], xs[1:]
        torch.cuda.empty_cache()
        distill = leg(distill_steps_per_sec, dev, rank, n_gpus, distributed)
        if n_gpus == 1 and not args.no_diffquant:
            torch.cuda.empty_cache()
            distill['diffquant_wrn'] = leg(diffquant_steps_per_sec, dev)
        # configs[3] is quoted on 8 GPUs, configs[4] on 4: run them where BASELINE.json places them
        if n_gpus == 8 or os.environ.get('QD_BENCH_CFG4') == '1':
            distill['imagenet_resnet18k_dp'] = leg(dp_config_steps_per_sec, 'imagenet', dev, rank, n_gpus, distributed)
        if n_gpus == 4 or os.environ.get('QD_BENCH_CFG5') == '1':
            distill['nmt_lstm_dp'] = leg(dp_config_steps_per_sec, 'nmt', dev, rank, n_gpus, distributed)

    # RCCL writes a version banner to the C-level stdout, which is block-buffered when piped and would
    # otherwise be flushed at exit, i.e. after the JSON line: push it out now on every rank, so that the JSON
    # is the last thing on stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if distributed:
        dist.barrier()
    if rank == 0:
        bytes_per_launch = ALGO_BYTES_PER_ELEM * N_ELEM
        total_bytes = bytes_per_launch * args.steps * n_gpus
        value = total_bytes / elapsed / 1e9
        kernel_us = event_ms * 1e3 / args.steps
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        out = {
            'metric': 'quantize_dequantize_GBps_64M_fp32_4bit',
            'value': round(value, 2), 'unit': 'GB/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed * 1e3 / args.steps, 5), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'quantization.uniformQuantization(x, s=16, type_of_scaling="linear", bucket_size=256), '
                            'x = randn(64Mi) fp32 per GPU, deterministic rounding (BASELINE configs[1] hot path at the '
                            "metric's headline size)",
                'n_elements_per_gpu': N_ELEM, 'levels': LEVELS, 'bucket_size': BUCKET,
                'algorithmic_bytes_per_element': ALGO_BYTES_PER_ELEM, 'rotating_buffers': N_ROTATE, 'precondition_launches': n_pre,
                'parallelism': 'independent tensors per rank (no data-path collective)' if n_gpus > 1 else 'single GPU',
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'k_bucket_vec<MODE_QDQ,16,4>',
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBPS, 4), 'traffic': load_pmc_traffic(),
                'avg_launch_us': round(kernel_us, 3), 'algorithmic_bytes_per_launch': bytes_per_launch,
                'timing': 'HIP events on the launch stream around the %d timed launches (includes inter-launch gaps)' % args.steps,
                'torch_d2d_copy_GBps': round(copy_gbps, 1),      # torch's own copy of the same bytes, same box (a hand-written NT copy reaches 6.3-6.5 TB/s: profiles/r01_kbench.txt)
            },
            'cpu_baseline': cpu,
            'distill': distill,
            'parity_bit_exact_vs_oracle': parity,
            'device': torch.cuda.get_device_name(dev),
        }
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
