#!/usr/bin/env python3
"""initialize_quantization_points on model-sized tensors: where the time goes (scale, sort, gather)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
import quantization.help_functions as qhf  # noqa: E402


def wall(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (800000, 5308416, 17842176):
    x = torch.randn(n, device='cuda') * 0.05
    sf = quantization.ScalingFunction('linear', False, False, 256, False)
    u = sf.scale_down(x).view(-1)[:n]
    print('n = %9d: scale_down %8.1f us   torch.sort %8.1f us   initialize_quantization_points(k=4) %8.1f us   (k=16) %8.1f us'
          % (n, wall(lambda: sf.scale_down(x)), wall(lambda: torch.sort(u)),
             wall(lambda: qhf.initialize_quantization_points(x, sf, 4)), wall(lambda: qhf.initialize_quantization_points(x, sf, 16))), flush=True)

import numpy as np  # noqa: E402

print('order_statistics alone (8 ranks / 32 ranks / 64 ranks) vs torch.sort:')
for n in (800000, 5308416, 17842176):
    u = torch.rand(n, device='cuda')
    rng = np.random.default_rng(0)
    row = []
    for m in (8, 32, 64):
        ranks = np.sort(rng.integers(0, n, size=m))
        row.append(wall(lambda: qhf.order_statistics(u, ranks)))
    print('n = %9d: %8.1f %8.1f %8.1f us   sort %8.1f us' % (n, row[0], row[1], row[2], wall(lambda: torch.sort(u))), flush=True)
