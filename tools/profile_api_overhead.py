#!/usr/bin/env python3
"""Host-side cost of one quantization.uniformQuantization call on a tiny tensor (launch-bound
regime: 13 of the CIFAR student's 22 tensors have < 256 elements)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

x = torch.randn(500, device='cuda')
big = torch.randn(800000, device='cuda')
for _ in range(200):
    quantization.uniformQuantization(x, 16, bucket_size=256)
torch.cuda.synchronize()
for name, t in (('500 elements', x), ('800000 elements', big)):
    t0 = time.perf_counter()
    for _ in range(5000):
        quantization.uniformQuantization(t, 16, bucket_size=256)
    torch.cuda.synchronize()
    print('%-16s %.2f us per call (wall, 5000 calls)' % (name, (time.perf_counter() - t0) / 5000 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    quantization.uniformQuantization(x, 16, bucket_size=256)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
