#!/usr/bin/env python3
"""A few dispatches of the kernels round 2 tunes, for SQ-counter passes (rocprofv3 --pmc ...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import codec  # noqa: E402

N = 1 << 26
xs = [torch.randn(N, device='cuda') for _ in range(3)]
keep = []
for b in (256, 100, 33, 250):
    for i in range(3):
        keep.append(quantization.uniformQuantization(xs[i], 16, bucket_size=b)[0])
    torch.cuda.synchronize()
    keep.clear()
g = torch.randn(N, device='cuda')
for k in (16, 256):
    pts = torch.sort(torch.rand(k, device='cuda'))[0]
    fn = quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=xs[0])
    fn.forward(None, pts)
    for i in range(3):
        fn.backward(g)
    torch.cuda.synchronize()
lev = torch.randint(0, 16, (N,), dtype=torch.uint8, device='cuda')
for i in range(3):
    codec.histogram_u8(lev, 16)
torch.cuda.synchronize()
print('ok')
