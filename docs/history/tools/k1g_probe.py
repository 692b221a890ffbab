#!/usr/bin/env python3
"""Un-bucketed (bucket_size=None) quantization of model-sized tensors: time per API call with the one-launch
register-resident kernel (k_single_fused) and with the three-launch path (reduce, fold, apply), same process.
`--pmc` mode: a few launches per size and path only, for a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import _lib  # noqa: E402
from quantized_distillation_amd.multi_tensor import MultiTensorQuantizer  # noqa: E402

lib = _lib.load()
SIZES = (100000, 800000, 2841600, 5308416, 17842176)
if '--pmc' in sys.argv:
    for n in (800000, 5308416, 17842176):
        xs = [torch.randn(n, device='cuda') for _ in range(4)]
        keep = []
        for mode in (1, 0):
            lib.qd_set_single_fused_mode(mode)
            for i in range(4):
                keep.append(quantization.uniformQuantization(xs[i], 16)[0])
            torch.cuda.synchronize()
    print('pmc launches done')
    sys.exit(0)

print('%-22s %12s %12s %14s' % ('n (MB)', 'fused us', '3-launch us', 'fused GB/s @8B'))
for n in SIZES:
    xs = [torch.randn(n, device='cuda') for _ in range(3)]
    live = [None] * 3

    def step(i):
        live[i % 3] = quantization.uniformQuantization(xs[i % 3], 16)[0]
    res = {}
    for mode in (1, 0):
        lib.qd_set_single_fused_mode(mode)
        for i in range(200):
            step(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for r in range(3):
            torch.cuda.synchronize()
            e0.record()
            for i in range(200):
                step(i)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        res[mode] = best
    print('%9d (%5.1f MB) %12.2f %12.2f %14.0f' % (n, n * 4 / 1e6, res[1], res[0], 8 * n / res[1] / 1e3))
lib.qd_set_single_fused_mode(-1)
