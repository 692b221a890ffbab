#!/usr/bin/env python3
"""configs[1] step (ConvolForwardNet student, 4-bit, bucket 256, batch 50): eager multi-tensor step against the same step
replayed from hipGraphs (DistillTrainer.capture), interleaved repetitions of 100 steps each."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from harness import models  # noqa: E402
from harness.distill import DistillTrainer, synthetic_batch  # noqa: E402

dev = torch.device('cuda:0')
batches = [synthetic_batch(50, dev, seed=i) for i in range(4)]
tr = {}
for name in ('eager', 'graph'):
    torch.manual_seed(0)
    tr[name] = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode='multi')
    for i in range(20):
        tr[name].step(*batches[i % 4])
tr['graph'].capture(*batches[0])
for i in range(20):
    tr['graph'].step(*batches[i % 4])
reps = {k: [] for k in tr}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 9):
    for name in tr:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            tr[name].step(*batches[i % 4])
        torch.cuda.synchronize()
        reps[name].append(100 / (time.perf_counter() - t0))
for name, v in reps.items():
    print('%-6s steps/s: median %.1f  min %.1f  max %.1f   %s' % (name, statistics.median(v), min(v), max(v), ' '.join('%.0f' % x for x in v)))
# the two must train the same model: compare the parameters after the same number of steps from the same seed
pe = torch.cat([p.detach().view(-1) for p in tr['eager'].student.parameters()])
pg = torch.cat([p.detach().view(-1) for p in tr['graph'].student.parameters()])
print('max |param difference| eager vs graph after the same steps: %.3g (max |param| %.3g)' % (float((pe - pg).abs().max()), float(pe.abs().max())))
