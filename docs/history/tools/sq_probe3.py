#!/usr/bin/env python3
"""A few dispatches of the kernels the closing session of round 2 added or changed, for SQ-counter passes
(rocprofv3 --pmc ...): the one-wave-per-bucket kernel at bucket 1000 / 513 / 5000 next to the vector kernel at 256 / 1024,
the chunk_any kernel at 33, the LDS-atomic histogram at k = 16 / 256."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import codec  # noqa: E402

N = 1 << 26
xs = [torch.randn(N, device='cuda') for _ in range(3)]
keep = []
for b in (256, 1024, 1000, 513, 5000, 33):
    for i in range(3):
        keep.append(quantization.uniformQuantization(xs[i], 16, bucket_size=b)[0])
    torch.cuda.synchronize()
    keep.clear()
for k in (16, 256):
    lev = [torch.randint(0, k, (N,), dtype=torch.uint8, device='cuda') for _ in range(3)]
    for i in range(3):
        codec.histogram_u8(lev[i], k)
    torch.cuda.synchronize()
print('ok')
