#!/usr/bin/env python3
"""Cost of a tensor view that starts 4 bytes into a 16-byte granule (x[1:]) against an aligned tensor, 64 Mi elements."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

dev = 'cuda:0'
N = 1 << 26
base = [torch.randn(N + 8, device=dev) for _ in range(3)]
pts = torch.tensor([0.0, 0.3, 0.6, 1.0], device=dev)


def t(name, fn, bpe):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print('%-70s %8.2f us %5.1f%%' % (name, best, bpe * N / best / 1e3 / 80), flush=True)


keep = [None] * 3
for off in (0, 1):
    xs = [b[off:off + N] for b in base]
    tag = 'aligned' if off == 0 else 'view at +4 bytes'
    for bk in (256, 1000, 33, None):
        t('K1 uniform bucket %s, %s' % (bk, tag), lambda i: keep.__setitem__(i % 3, quantization.uniformQuantization(xs[i % 3], 16, bucket_size=bk)[0]), 8 if bk else 12)
    sf = quantization.ScalingFunction('linear', False, False, 256)
    t('K2 scale_down bucket 256, %s' % tag, lambda i: keep.__setitem__(i % 3, sf.scale_down(xs[i % 3])), 8)
    t('K4 nonUniform k=4 bucket 256, %s' % tag, lambda i: keep.__setitem__(i % 3, quantization.nonUniformQuantization(xs[i % 3], pts, bucket_size=256)[0]), 16)
