#!/usr/bin/env python3
"""A/B timing of the headline kernel under different launch geometries (QD_GRID_CAP), next to
torch's device copy of the same bytes.  Interleaved rounds in one process; prints a table.
Needs a library built with -DQD_TUNING (the QD_GRID_CAP hook is compiled out of the product build).
Run on the GPU box:  python tools/tune_k1.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

N = 64 * 1024 * 1024
dev = torch.device('cuda:0')
xs = [torch.randn(N, device=dev) for _ in range(4)]
outs = [torch.empty(N, device=dev) for _ in range(4)]


def timeit(fn, iters=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def quant(i):
    q, _ = quantization.uniformQuantization(xs[i % 4], 16, bucket_size=256)
    outs[i % 4] = q


def copy(i):
    outs[i % 4].copy_(xs[i % 4])


caps = [512, 1024, 2048, 4096, 8192, 16384, 65536]
res = {c: [] for c in caps}
res['copy'] = []
for rnd in range(5):
    for c in caps:
        os.environ['QD_GRID_CAP'] = str(c)
        res[c].append(timeit(quant))
    res['copy'].append(timeit(copy))
os.environ.pop('QD_GRID_CAP', None)
for k, v in res.items():
    v = sorted(v)
    print('%8s  min %8.2f us  med %8.2f us   %7.1f GB/s (8 B/elem, at min)' % (k, v[0], v[len(v) // 2], 8 * N / v[0] / 1e3))
