"""A rank of tests/test_legs_gloo.py: the shape of bench.py's leg sequence on two gloo ranks (CPU).  Leg 'breaks' raises on
rank 1 BEFORE its all-reduce while rank 0 is already inside it; leg 'dp' is a toy data-parallel step reported through
harness.dpbench.dp_report.  Rank 0 prints one JSON line."""
import datetime
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from harness import legs  # noqa: E402
from harness.dpbench import dp_report, flat_dp  # noqa: E402
from harness.flat import GradSynchronizer  # noqa: E402

t_start = time.time()
dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=4))        # the data-path group: a short timeout
rank, world = dist.get_rank(), dist.get_world_size()
runner = legs.LegRunner(ctl_timeout_s=60, log=lambda s: None)
dev = torch.device('cpu')
out = {}


def fine():
    t = torch.ones(4)
    dist.all_reduce(t)
    return {'sum': float(t[0])}


def dp():
    torch.manual_seed(0)
    w = torch.randn(64, 64)
    grad = torch.zeros(1 << 16)
    sync = GradSynchronizer(grad, chunks=2)

    def step(i):
        grad.copy_((w @ w).sum() * torch.ones_like(grad))
        sync.sync()

    def set_exchange(on):
        sync.active = on and sync.world_active
    return dp_report(step, 3, 3, dev, world, True, 8, grad.numel() * 4, set_exchange, sync.sync, runner.barrier, rank)


def breaks():
    if rank == 1:
        raise RuntimeError('rank 1 ran out of memory (simulated)')
    t = torch.ones(4)
    dist.all_reduce(t)                  # rank 0 waits here for a rank that never comes: the group's timeout ends it
    return {'sum': float(t[0])}


def local():
    return {'rank': rank}


out['fine'] = runner.run('fine', fine)
out['dp'] = runner.run('dp', dp)
out['dp_flat'] = flat_dp(out['dp'])
out['breaks'] = runner.run('breaks', breaks)
out['after'] = runner.run('after', fine)                       # holds collectives: must be skipped, not attempted
out['local'] = runner.run('local', local, collective=False)    # no collectives: still runs
out['history'] = runner.history
out['seconds'] = round(time.time() - t_start, 2)
runner.barrier()
if rank == 0:
    print(json.dumps(out), flush=True)
os._exit(0)                                                    # (as bench.py does when the communicator is broken)
