#!/usr/bin/env python3
"""A/B of nonUniformQuantization from raw x (K4, int64 indices) at bucket sizes off the vector path: the library built with
-DQD_RAW_TILES=0 (k_bucket_chunk_any: three LDS phases per chunk) against -DQD_RAW_TILES=1 (k_nearest_raw_tiles: the tile
in registers, min / max through integer LDS atomics).  Same box, HIP events, 3 rotating tensors; both must give the same bits.
    python tools/ab_k4.py            (expects build/ab/libqd_rawtiles_{0,1}.so; builds them when missing)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from harness.kernel_bench import time_row  # noqa: E402
from quantized_distillation_amd import _lib, build  # noqa: E402

out_dir = os.path.join(ROOT, 'build', 'ab')
os.makedirs(out_dir, exist_ok=True)
libs = {}
for v in (0, 1):
    so = os.path.join(out_dir, 'libqd_rawtiles_%d.so' % v)
    if not os.path.exists(so):
        subprocess.check_call([build.hipcc()] + build.HIPCC_FLAGS + ['-I', _lib.INCLUDE, '-DQD_RAW_TILES=%d' % v] +
                              [os.path.join(_lib.CSRC, f) for f in build.SOURCES] + ['-o', so])
    lib = ctypes.CDLL(so)
    lib.qd_nearest_point_f32.restype = ctypes.c_int
    lib.qd_nearest_point_f32.argtypes = _lib.SIGNATURES['qd_nearest_point_f32'][1]
    lib.qd_workspace_bytes.restype = ctypes.c_size_t
    libs[v] = lib
dev = torch.device('cuda:0')
N = 1 << 26
xs = [torch.randn(N, device=dev) for _ in range(3)]
qs = [torch.empty(N, device=dev) for _ in range(3)]
ids = [torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3)]
ws = torch.empty(libs[0].qd_workspace_bytes(), dtype=torch.uint8, device=dev)
for bucket in (33, 50, 250, 7, 255, 447, 100, 256):
    nb = -(-N // bucket)
    ab = torch.empty(2, nb, device=dev)
    for k in (4, 16):
        pts = torch.sort(torch.rand(k, device=dev))[0]
        res = {}
        for v in (0, 1):
            fn = libs[v].qd_nearest_point_f32

            def call(i):
                j = i % 3
                rc = fn(xs[j].data_ptr(), 0, pts.data_ptr(), k, 0, qs[j].data_ptr(), ids[j].data_ptr(), 8, N, bucket, ab[0].data_ptr(),
                        ab[1].data_ptr(), None, 0, 0.0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
                assert rc == 0, rc
            us, lo, hi = time_row(call, iters=20)
            call(0)
            torch.cuda.synchronize()
            res[v] = (us, qs[0].clone(), ids[0].clone(), ab.clone())
        same = all(torch.equal(a, b) for a, b in zip(res[0][1:], res[1][1:]))
        print('bucket %4d k %2d: chunk_any %7.2f us (%.1f%%)   raw_tiles %7.2f us (%.1f%%)   same bits: %s'
              % (bucket, k, res[0][0], 16 * N / res[0][0] / 1e3 / 80, res[1][0], 16 * N / res[1][0] / 1e3 / 80, same), flush=True)
