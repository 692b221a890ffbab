#!/usr/bin/env python3
"""Runs the bucketed uniform kernel at bucket 256 and 64 (N = 64 Mi) a few times for an SQ-counter pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402

N = 1 << 26
xs = [torch.randn(N, device='cuda') for _ in range(3)]
keep = []
for b in (256, 64, 1024, 128):
    for i in range(3):
        keep.append(quantization.uniformQuantization(xs[i], 16, bucket_size=b)[0])
    torch.cuda.synchronize()
    keep.clear()
print('ok')
