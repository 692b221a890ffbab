#!/usr/bin/env python3
"""Does the order of the two legs (or the persistent flat shadows) explain multi < per_tensor in the bench?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from harness import models  # noqa: E402
from harness.distill import DistillTrainer, synthetic_batch  # noqa: E402

dev = torch.device('cuda:0')
if '--rccl' in sys.argv:
    import torch.distributed as dist
    from harness import launch
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % launch.free_port(), rank=0, world_size=1, device_id=dev)
    os.environ['QD_FORCE_DIST'] = '1'
batches = [synthetic_batch(50, dev, seed=i) for i in range(4)]
for mode in ('multi', 'per_tensor', 'multi', 'per_tensor'):
    torch.manual_seed(0)
    tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode=mode)
    for i in range(40):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(100):
            tr.step(*batches[i % 4])
        torch.cuda.synchronize()
        res.append(100 / (time.perf_counter() - t0))
    print('%-10s steps/s %s' % (mode, ' '.join('%.1f' % r for r in res)), flush=True)
    del tr
