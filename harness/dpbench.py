"""Timing of a data-parallel training step and the quantities its design is judged on.

bench.py's steps/sec legs (BASELINE configs[1..4]) all report through dp_report(): steps/sec as
the median of several repetitions (max over ranks per repetition), the per-rank spread, the
gradient exchange on its own (time, ring bus bandwidth next to the xGMI peak), what the exchange
costs the step (exposed communication), and the efficiency against the same step on ONE GPU,
measured in the same run.  Device-agnostic: on a CPU device the HIP events become the host clock
and the synchronisations disappear, so tests/test_legs_gloo.py runs it on two gloo ranks.
"""
import statistics
import time

import torch
import torch.distributed as dist

XGMI_PEAK_GBPS = 7 * 153.0         # per GPU: 7 point-to-point xGMI links x ~153 GB/s (MI355X_MICROARCH.md)


def _sync(dev):
    if torch.device(dev).type == 'cuda':
        torch.cuda.synchronize()


def event_ms(fn, iters, precondition_s=0.1, reps=3, dev='cuda'):
    """Median over `reps` of the time of `iters` back-to-back calls of fn(), in ms per call, after `precondition_s`
    seconds of untimed calls.  On a HIP device: HIP events on the launch stream, harness/kernel_bench.py's method -- the
    phase figures of the steps/sec legs are measured like the kernel rows, so the two agree."""
    if torch.device(dev).type == 'cuda':
        from .kernel_bench import time_row
        us, _lo, _hi = time_row(lambda _i: fn(), iters=iters, reps=reps, precondition_s=precondition_s)
        return us / 1e3
    samples = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _i in range(iters):
            fn()
        samples.append((time.perf_counter() - t0) / iters * 1e3)
    return statistics.median(samples)


def timed_steps(step, steps, reps, dev, distributed):
    """`reps` repetitions of `steps` calls of step(i), each bracketed by synchronize (+ barrier + synchronize when
    distributed).  Returns (job seconds per repetition = max over ranks, this rank's own seconds per repetition)."""
    job, own = [], []
    for _ in range(reps):
        _sync(dev)
        if distributed:
            dist.barrier()
            _sync(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        _sync(dev)
        mine = time.perf_counter() - t0
        if distributed:
            dist.barrier()
            _sync(dev)
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        job.append(dt)
        own.append(mine)
    return job, own


def dp_report(step, steps, reps, dev, n_gpus, distributed, per_gpu_batch, grad_bytes, set_exchange, allreduce_once, ctl_barrier, rank):
    """Times `step` data parallel and fills the quantities the data-parallel design is judged on:
      steps_per_sec (median of `reps` repetitions, with min / max), per-rank step time min / max,
      allreduce_alone_ms  the step's gradient exchange on its own, un-overlapped, event timed,
      busbw_GBps          2 (N-1)/N x bytes / that time (the ring bus bandwidth), next to the xGMI peak per GPU,
      exposed_comm_ms     step time minus the time of the same step with the exchange switched off,
      dp_efficiency       time of the step on ONE GPU (rank 0 alone, the others idle, no exchange) / data-parallel step time.
    set_exchange(bool) switches the step's exchange off and on; allreduce_once() issues the step's exchange once (None: the
    step has none); ctl_barrier(ok) -> list of ranks that said not-ok, must not touch the data-path communicator
    (harness/legs.py LegRunner.barrier)."""
    job, own = timed_steps(step, steps, reps, dev, distributed)
    ms = [t / steps * 1e3 for t in job]
    med = statistics.median(ms)
    out = {'steps_per_sec': round(1e3 / med, 3), 'ms_per_step': round(med, 3), 'statistic': 'median of %d repetitions of %d steps' % (reps, steps),
           'steps_per_sec_min': round(1e3 / max(ms), 3), 'steps_per_sec_max': round(1e3 / min(ms), 3),
           'ms_per_step_repetitions': [round(m, 3) for m in ms],
           'samples_per_sec': round(per_gpu_batch * n_gpus * 1e3 / med, 1), 'n_gpus': n_gpus,
           'per_gpu_batch': per_gpu_batch, 'global_batch': per_gpu_batch * n_gpus, 'steps': steps}
    # this rank's own step time, best repetition; min / max over the ranks
    mine = min(own) / steps * 1e3
    if distributed:
        t = torch.tensor([mine, -mine], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out['rank_ms_per_step'] = {'max': round(float(t[0]), 3), 'min': round(-float(t[1]), 3)}
    else:
        out['rank_ms_per_step'] = {'max': round(mine, 3), 'min': round(mine, 3)}
    # the exchange on its own
    out['exchanged_bytes_per_step'] = int(grad_bytes)
    if allreduce_once is not None:
        ar = event_ms(allreduce_once, iters=10, precondition_s=0.02, dev=dev)
        if distributed:
            t = torch.tensor([ar], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ar = float(t[0])
        out['allreduce_alone_ms'] = round(ar, 4)
        out['busbw_GBps'] = round(2.0 * (n_gpus - 1) / n_gpus * grad_bytes / (ar * 1e-3) / 1e9, 1) if n_gpus > 1 else 0.0
        out['algbw_GBps'] = round(grad_bytes / (ar * 1e-3) / 1e9, 1)
        out['xgmi_peak_GBps_per_gpu'] = XGMI_PEAK_GBPS
    # the same step without the exchange, all ranks at once: what the exchange costs the step
    set_exchange(False)
    try:
        job0, _own0 = timed_steps(step, steps, reps, dev, distributed)
        nocomm = statistics.median(t / steps * 1e3 for t in job0)
        out['ms_per_step_without_exchange'] = round(nocomm, 3)
        out['exposed_comm_ms'] = round(med - nocomm, 3)
        # ... and on ONE GPU with the others idle: the N = 1 time of this very leg, measured in this run
        if distributed:
            n1, solo_err = None, None
            if rank == 0:
                try:
                    j1, _ = timed_steps(step, steps, reps, dev, False)
                    n1 = statistics.median(t / steps * 1e3 for t in j1)
                except Exception as e:                  # noqa: BLE001 -- told to the others below, then raised everywhere
                    solo_err = e
            # The other ranks wait here (gloo), off the GPUs.  The wait CARRIES rank 0's verdict: if its solo timing raised,
            # every rank raises here, before the data-path broadcast below -- otherwise rank 0's end-of-leg agreement would
            # pair with this in-leg one, the others would sit in the broadcast until the data timeout, and every later
            # agreement would be off by one.
            failed = ctl_barrier(solo_err is None) or []
            if failed:
                raise RuntimeError('the one-GPU-alone timing failed on rank(s) %s%s' % (failed, ': %s' % solo_err if solo_err else ''))
            t = torch.tensor([n1 if n1 is not None else 0.0], dtype=torch.float64, device=dev)
            dist.broadcast(t, src=0)
            n1 = float(t[0])
        else:
            n1 = nocomm
    finally:
        set_exchange(True)
    out['ms_per_step_one_gpu_alone'] = round(n1, 3)
    out['dp_efficiency'] = round(n1 / med, 4)
    return out


DP_KEYS = ('steps_per_sec', 'steps_per_sec_min', 'steps_per_sec_max', 'ms_per_step', 'ms_per_step_repetitions', 'rank_ms_per_step',
           'exchanged_bytes_per_step', 'allreduce_alone_ms', 'busbw_GBps', 'algbw_GBps', 'xgmi_peak_GBps_per_gpu',
           'ms_per_step_without_exchange', 'exposed_comm_ms', 'ms_per_step_one_gpu_alone', 'dp_efficiency')


def flat_dp(d):
    """One steps/sec leg as a short string for the driver's record (only scalars survive there)."""
    if not isinstance(d, dict) or 'steps_per_sec' not in d:
        return str((d or {}).get('error') or (d or {}).get('skipped') or d)[:118]
    s = '%.2f steps/s (%.2f-%.2f) N=%d' % (d['steps_per_sec'], d['steps_per_sec_min'], d['steps_per_sec_max'], d['n_gpus'])
    if 'allreduce_alone_ms' in d:
        s += ' | ar %.3f ms busbw %.0f/%.0f GB/s' % (d['allreduce_alone_ms'], d['busbw_GBps'], XGMI_PEAK_GBPS)
    s += ' | exposed %.3f ms | eff %.3f | rank ms %.2f-%.2f' % (d['exposed_comm_ms'], d['dp_efficiency'],
                                                               d['rank_ms_per_step']['min'], d['rank_ms_per_step']['max'])
    return s[:118]
