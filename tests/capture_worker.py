"""Child process of tests/test_hip_capture_watchdog.py: the situation round 4's driver run died in.

A one-rank RCCL process group (so the c10d watchdog thread exists), QD_FORCE_DIST=1 (so every training step issues a real
all-reduce), >= 20 eager steps, then DistillTrainer.capture() IMMEDIATELY, `--iters` times over.  A separate process because
the failure mode is std::terminate on the watchdog thread -- SIGABRT, which no pytest process survives.

    --mode thread_local|global    capture error mode (the product uses thread_local; 'global' is what round 4 ran)
    --settle S                    seconds quiesce_collectives() gives the watchdog to retire finished work (product: 0.35)
    --inflight K                  K un-waited async all-reduces issued right before the capture starts: the watchdog is
                                  GUARANTEED to be polling completion events while the capture runs
    --long-capture S              instead of the trainer loop: ONE capture that stays open for S seconds (a time-based loop of tiny
                                  launches) while an all-reduce is kept incomplete behind a 1.5 x S s spin kernel -- the watchdog
                                  (100 ms period) polls that work item's event several times while the capture is open.  With
                                  --mode global this is round 4's crash on demand; with thread_local it must pass
    --failed-capture HOW          instead: a capture whose body fails (raise: a Python exception; sync: a device synchronize,
                                  which invalidates the capture; item: a blocking copy) must restore the caller's stream, leave the process
                                  usable and the trainer eager (its own process too: a botched capture must not be able to
                                  take the rest of the GPU suite with it)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ['QD_FORCE_DIST'] = '1'

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from harness import launch, models  # noqa: E402
from harness.distill import DistillTrainer, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--mode', default='thread_local')
ap.add_argument('--settle', type=float, default=0.35)
ap.add_argument('--inflight', type=int, default=0)
ap.add_argument('--failed-capture', default=None, choices=['raise', 'sync', 'item'], help='the failure path of capture_into() instead')
ap.add_argument('--long-capture', type=float, default=0.0, help='deterministic provocation: see the module docstring')
args = ap.parse_args()

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % launch.free_port(), rank=0, world_size=1, device_id=dev)
batches = [synthetic_batch(50, dev, seed=i) for i in range(4)]
scratch = [torch.ones(1 << 20, device=dev) for _ in range(max(1, args.inflight))]
handles = []


def inflight():
    for t in scratch[:args.inflight]:
        handles.append(dist.all_reduce(t, async_op=True))       # not waited for: the watchdog lists them


def mark(s):
    print(s, flush=True)


if args.long_capture > 0:
    import time
    from harness.distill import capture_into
    x = torch.ones(1024, device=dev)
    y = torch.ones(1 << 20, device=dev)
    dist.all_reduce(y)
    torch.cuda.synchronize()
    # cycles for ~1.5 x S seconds of spinning (the shader clock is ~2.4 GHz; measured, not assumed)
    t0 = time.perf_counter()
    torch.cuda._sleep(200_000_000)
    torch.cuda.synchronize()
    per_cycle = (time.perf_counter() - t0) / 200_000_000
    torch.cuda._sleep(int(1.5 * args.long_capture / per_cycle))          # on the current stream ...
    h = dist.all_reduce(y, async_op=True)                               # ... which the RCCL stream waits for: incomplete for ~1.5 S
    g = torch.cuda.CUDAGraph()
    n = [0]

    def body():
        t_end = time.time() + args.long_capture
        while time.time() < t_end:                  # ~25 nodes: the capture is LONG, not large
            x.add_(1)
            n[0] += 1
            time.sleep(0.02)
    mark('capturing for %.2f s in %s mode with an incomplete all-reduce listed by the watchdog' % (args.long_capture, args.mode))
    capture_into(g, body, stream=torch.cuda.Stream(), error_mode=args.mode)
    h.wait()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert float(x[0]) == 1 + n[0], (float(x[0]), n[0])
    mark('LONG_CAPTURE_OK %d launches captured' % n[0])
    dist.destroy_process_group()
    sys.exit(0)

if args.failed_capture:
    from harness.distill import capture_into
    before = torch.cuda.current_stream()
    g = torch.cuda.CUDAGraph()
    x = torch.ones(1024, device=dev)

    def bad():
        x.add_(1)
        if args.failed_capture == 'raise':
            raise ValueError('the captured body raised')
        if args.failed_capture == 'sync':
            torch.cuda.synchronize()        # illegal inside a capture: invalidates it, capture_end() then raises too
        else:
            x.sum().item()                  # a blocking device-to-host copy

    mark('capturing a body that fails (%s)' % args.failed_capture)
    try:
        capture_into(g, bad, stream=torch.cuda.Stream())
        raise SystemExit('the capture of a failing body did not fail')
    except SystemExit:
        raise
    except Exception as e:                  # noqa: BLE001
        mark('capture failed as it should: %s: %s' % (type(e).__name__, str(e).splitlines()[0][:150]))
    assert torch.cuda.current_stream() == before, 'the stream was not restored'
    mark('stream restored')
    y = (torch.arange(8, device=dev) * 2).sum()
    torch.cuda.synchronize()
    assert int(y) == 56
    mark('eager launches work')
    torch.manual_seed(0)
    tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi')
    tr.step(*batches[0])

    def boom():
        raise RuntimeError('fails between quiesce and capture')

    try:
        tr.capture(*batches[0], _before_capture=boom)
        raise SystemExit('capture() swallowed the error')
    except RuntimeError:
        pass
    assert tr._graph_fb is None and torch.cuda.current_stream() == before
    mark('trainer still eager')
    loss = tr.step(*batches[0])             # still an eager trainer, still working
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    mark('eager step after the failed capture works')
    stage = 'simple,fresh'
    if 'simple' in stage:                   # a fresh, simple capture in the same process
        g2 = torch.cuda.CUDAGraph()
        capture_into(g2, lambda: x.add_(1), stream=torch.cuda.Stream())
        x0 = float(x[0])
        g2.replay()
        torch.cuda.synchronize()
        assert float(x[0]) == x0 + 1
        mark('a later simple capture works')
    if 'fresh' in stage:                    # a fresh trainer captures and replays
        torch.manual_seed(0)
        t2 = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi')
        for i in range(3):
            t2.step(*batches[i])
        t2.capture(*batches[0])
        loss = t2.step(*batches[1])
        torch.cuda.synchronize()
        assert bool(torch.isfinite(loss)) and t2._graph_fb is not None
        mark('a fresh trainer captures and replays')
    try:                                    # the trainer whose capture failed refuses another attempt (it would segfault)
        tr.capture(*batches[0])
        raise SystemExit('a second capture attempt on the failed trainer was not refused')
    except RuntimeError as e:
        assert 'discard' in str(e)
    loss = tr.step(*batches[2])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and tr._graph_fb is None
    mark('the failed trainer refuses a second capture and keeps stepping eagerly')
    print('FAILED_CAPTURE_OK', flush=True)
    dist.destroy_process_group()
    sys.exit(0)

for it in range(args.iters):
    torch.manual_seed(0)
    tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi')
    assert tr.sync.active, 'QD_FORCE_DIST=1: the one-rank group must carry the all-reduces'
    for i in range(args.steps):
        tr.step(*batches[i % 4])                                 # eager, one all-reduce each
    assert tr.sync.collectives_issued >= args.steps
    before = torch.cuda.current_stream()
    tr.capture(*batches[0], error_mode=args.mode, settle_s=args.settle, _before_capture=inflight if args.inflight else None)
    assert torch.cuda.current_stream() == before, 'capture() must leave the current stream as it found it'
    for i in range(5):
        loss = tr.step(*batches[i % 4])                          # graph A, eager all-reduce, graph B
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)), 'replayed step produced a non-finite loss'
    for h in handles:
        h.wait()
    del handles[:]
    del tr
torch.cuda.synchronize()
print('CAPTURE_OK %d' % args.iters, flush=True)
dist.destroy_process_group()
