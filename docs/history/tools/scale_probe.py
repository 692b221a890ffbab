#!/usr/bin/env python3
"""Is the headline kernel's speed data dependent?  Same kernel, inputs randn * scale (+ offset)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

N = 1 << 26
dev = 'cuda:0'
base = [torch.randn(N, device=dev) for _ in range(4)]
live = [None] * 4


def t(xs, iters=60):
    def step(i):
        live[i % 4] = quantization.uniformQuantization(xs[i % 4], 16, bucket_size=256)[0]
    for i in range(10):
        step(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters):
        step(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for i in range(2000):
    quantization.uniformQuantization(base[i % 4], 16, bucket_size=256)
for rnd in range(2):
    for scale, off in [(1.0, 0.0), (0.05, 0.0), (0.001, 0.0), (20.0, 0.0), (1e-20, 0.0), (1.0, 100.0), (0.05, 1.0), (1.0, 0.0)]:
        xs = [b * scale + off for b in base]
        print('scale %-7g offset %-5g : %.2f us' % (scale, off, t(xs)))
        del xs
