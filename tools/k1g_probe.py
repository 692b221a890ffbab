#!/usr/bin/env python3
"""Un-bucketed (bucket_size=None) quantization of model-sized tensors: time per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

for n in (100000, 800000, 2841600, 5308416, 6000000, 17842176):
    xs = [torch.randn(n, device='cuda') for _ in range(3)]
    live = [None] * 3
    def step(i):
        live[i % 3] = quantization.uniformQuantization(xs[i % 3], 16)[0]
    for i in range(200):
        step(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for r in range(3):
        torch.cuda.synchronize(); e0.record()
        for i in range(200):
            step(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print('n = %9d (%5.1f MB): %7.2f us per call  -> %.0f GB/s on the 12 B/elem basis' % (n, n * 4 / 1e6, best, 12 * n / best / 1e3))
