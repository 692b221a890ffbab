#!/usr/bin/env python3
"""A/B: int64 index stores of the nearest-point vector kernel plain against non-temporal (-DQD_IDX_NT), with the index output
written to FRESH memory on every call (3 rotating 512 MB buffers) and to the same buffer on every call (what a caller whose
allocator hands the block back sees).  build/ab/libqd_idx_{plain,nt}.so; same box, HIP events."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from harness.kernel_bench import time_row  # noqa: E402
from quantized_distillation_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
N = 1 << 26
xs = [torch.randn(N, device=dev) for _ in range(3)]
qs = [torch.empty(N, device=dev) for _ in range(3)]
ids = [torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3)]
ab = torch.empty(2, N // 256, device=dev)
for tag in ('plain', 'nt'):
    lib = ctypes.CDLL(os.path.join(ROOT, 'build', 'ab', 'libqd_idx_%s.so' % tag))
    fn = lib.qd_nearest_point_f32
    fn.restype = ctypes.c_int
    fn.argtypes = _lib.SIGNATURES['qd_nearest_point_f32'][1]
    lib.qd_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(lib.qd_workspace_bytes(), dtype=torch.uint8, device=dev)
    for k in (4, 16):
        pts = torch.sort(torch.rand(k, device=dev))[0]
        for label, rot in (('fresh index buffer every call', 3), ('same index buffer every call', 1)):
            def call(i):
                j = i % 3
                rc = fn(xs[j].data_ptr(), 0, pts.data_ptr(), k, 0, qs[j].data_ptr(), ids[i % rot].data_ptr(), 8, N, 256, ab[0].data_ptr(),
                        ab[1].data_ptr(), None, 0, 0.0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
                assert rc == 0
            us, lo, hi = time_row(call, iters=30)
            print('%-5s k=%2d %-32s %7.2f us (%.2f..%.2f)  %.1f%% of 8 TB/s' % (tag, k, label, us, lo, hi, 16 * N / us / 1e3 / 80), flush=True)
