#!/usr/bin/env python3
"""Out-of-place vs in-place (modify_in_place=True) headline kernel: does writing over the input help HBM?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from quantized_distillation_amd import _lib  # noqa: E402

lib = _lib.load()
N = 1 << 26
xs = [torch.randn(N, device='cuda') for _ in range(8)]
outs = [torch.empty(N, device='cuda') for _ in range(4)]
ab = torch.empty(2, N // 256, device='cuda')


def run(i, inplace):
    x = xs[i % 8]
    o = x if inplace else outs[i % 4]
    lib.qd_uniform_f32(x.data_ptr(), o.data_ptr(), N, 256, 16, ab[0].data_ptr(), ab[1].data_ptr(), None, None, 0, 0.0, 0, 0,
                       None, 0, _lib.stream_ptr())


def t(inplace, iters=80):
    for i in range(30):
        run(i, inplace)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters):
        run(i, inplace)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for i in range(2000):
    run(i, False)
for r in range(3):
    print('out-of-place %.2f us   in-place %.2f us' % (t(False), t(True)))
