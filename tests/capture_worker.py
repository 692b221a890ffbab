"""Child process of tests/test_hip_capture_watchdog.py: the situation round 4's driver run died in.

A one-rank RCCL process group (so the c10d watchdog thread exists), QD_FORCE_DIST=1 (so every training step issues a real
all-reduce), >= 20 eager steps, then DistillTrainer.capture() IMMEDIATELY, `--iters` times over.  A separate process because
the failure mode is std::terminate on the watchdog thread -- SIGABRT, which no pytest process survives.

    --mode thread_local|global    capture error mode (the product uses thread_local; 'global' is what round 4 ran)
    --settle S                    seconds quiesce_collectives() gives the watchdog to retire finished work (product: 0.35)
    --inflight K                  K un-waited async all-reduces issued right before the capture starts: the watchdog is
                                  GUARANTEED to be polling completion events while the capture runs
    --failed-capture              instead: a capture whose body fails must restore the caller's stream, leave the process
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
ap.add_argument('--failed-capture', action='store_true', help='the failure path of capture_into() instead')
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


if args.failed_capture:
    from harness.distill import capture_into
    before = torch.cuda.current_stream()
    g = torch.cuda.CUDAGraph()
    x = torch.ones(1024, device=dev)

    def bad():
        x.add_(1)
        x.sum().item()                      # a synchronising call: illegal inside a capture

    try:
        capture_into(g, bad, stream=torch.cuda.Stream())
        raise SystemExit('the capture of a synchronising body did not fail')
    except SystemExit:
        raise
    except Exception as e:                  # noqa: BLE001
        print('capture failed as it should: %s' % type(e).__name__, flush=True)
    assert torch.cuda.current_stream() == before, 'the stream was not restored'
    y = (torch.arange(8, device=dev) * 2).sum()
    torch.cuda.synchronize()
    assert int(y) == 56
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
    loss = tr.step(*batches[0])             # still an eager trainer, still working
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    tr.capture(*batches[0])                 # and a later capture on the same trainer succeeds
    loss = tr.step(*batches[1])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and tr._graph_fb is not None
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
