"""Does the relative placement of input and output in HBM matter for the headline kernel?
Times qd_uniform_f32 (64 Mi fp32, 16 levels, bucket 256) with the output placed at input + 256 MiB + skew.
Usage (GPU box): python tools/offset_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantized_distillation_amd import _lib

dev = torch.device('cuda:0')
N = 64 << 20
lib = _lib.load()
R = 3
PAD = 64 << 20                                   # bytes of slack after every buffer
arena = torch.empty((2 * R) * (N * 4 + PAD) + (1 << 20), dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (1 << 20) - 1) & ~((1 << 20) - 1)      # 1 MiB aligned
stride = N * 4 + PAD
nb = N // 256
ab = torch.empty(2 * nb, device=dev)
src = torch.randn(N, device=dev)
st = _lib.stream_ptr(dev)


import ctypes
hip = ctypes.CDLL('libamdhip64.so')
xs = [base + (2 * r) * stride for r in range(R)]
for x in xs:
    hip.hipMemcpy(ctypes.c_void_p(x), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(N * 4), 3)
torch.cuda.synchronize()


def call(x, q):
    rc = lib.qd_uniform_f32(x, q, N, 256, 16, ab.data_ptr(), ab.data_ptr() + 4 * nb, None, None, 0, 0.0, 0, 0, None, 0, st)
    assert rc == 0


def timeit(skew, iters=60):
    qs = [base + (2 * r + 1) * stride + skew for r in range(R)]
    for i in range(200):
        call(xs[i % R], qs[i % R])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for i in range(iters):
            call(xs[i % R], qs[i % R])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for i in range(1500):
    call(xs[i % R], base + (2 * (i % R) + 1) * stride)
torch.cuda.synchronize()
for skew in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 3 << 20, 17 << 20, 0):
    print('output at input + %d MiB + %8d B : %7.2f us' % ((N * 4 + PAD) >> 20, skew, timeit(skew)), flush=True)
