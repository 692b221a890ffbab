#!/usr/bin/env python3
"""Cost of the ragged last bucket at large bucket sizes: N = 64 Mi + extra elements, so that the launch ends with a short
bucket of `extra` elements (plus, for the chunk kernels, the full buckets after the last whole chunk)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

dev = torch.device('cuda:0')
N = 1 << 26
for i in range(300):
    quantization.uniformQuantization(torch.empty(N, device=dev).normal_() if i == 0 else x0, 16, bucket_size=256) if i else None
    if i == 0:
        x0 = torch.randn(N, device=dev)
torch.cuda.synchronize()
for bucket, extra in ((256, 0), (256, 255), (1024, 1000), (4096, 0), (4096, 4000), (4096, 4001), (8192, 0), (8192, 8000), (8192, 8003), (2000, 1999),
                      (2000, 1), (8000, 7000), (8000, 7002), (1000, 0), (1000, 999), (513, 0), (513, 511)):
    n = N + extra if extra else N
    xs = [torch.randn(n, device=dev) for _ in range(3)]
    live = [None] * 3
    for i in range(60):
        live[i % 3] = quantization.uniformQuantization(xs[i % 3], 16, bucket_size=bucket)[0]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            live[i % 3] = quantization.uniformQuantization(xs[i % 3], 16, bucket_size=bucket)[0]
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 12 * 1e3)
    print('bucket %5d  N = 64Mi + %-5d %8.2f us  %5.1f%%' % (bucket, extra, best, 8 * n / best / 1e3 / 80), flush=True)
    del xs, live
