#!/usr/bin/env python3
"""scale_down (K2, qd_scale_down_f32) at N = 64 Mi for a few bucket sizes, HIP-event timing over rotating buffers.
QD_LIB=<path> loads another build of the library (same-box A/B, tools/build_rev_lib.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from quantized_distillation_amd import _lib  # noqa: E402

if os.environ.get('QD_LIB'):
    _lib.LIB_PATH = os.environ['QD_LIB']
lib = _lib.load()
N = 1 << 26
dev = torch.device('cuda:0')
R = 4
xs = [torch.randn(N, device=dev) for _ in range(R)]
ws = _lib.workspace(dev)
st = torch.cuda.current_stream().cuda_stream
for b in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '256,64,1024,100').split(',')]:
    padded = lib.qd_padded_length(N, b)
    us = [torch.empty(padded, device=dev) for _ in range(R)]
    nb = lib.qd_num_buckets(N, b)
    ab = torch.empty(2, nb, device=dev)

    def call(i):
        _lib.check(lib.qd_scale_down_f32(xs[i % R].data_ptr(), us[i % R].data_ptr(), N, b, ab[0].data_ptr(), ab[1].data_ptr(),
                                         None, 0, 0.0, ws.data_ptr(), ws.numel(), st))
    for i in range(20):
        call(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    print('K2 scale_down bucket %-5d %8.2f us  %7.1f GB/s  %4.1f%%' % (b, best, 8 * N / best / 1e3, 8 * N / best / 1e3 / 80))
