#!/usr/bin/env python3
"""The calls with a SIDE OUTPUT (point indices, level indices) at every bucket-size family of the kernels, N = 64 Mi:
    K4   nonUniformQuantization            x -> q + int64 indices            16 B/element
    K5   pre-processed forward (midpoint)   u -> q + uint8 indices             9 B/element
    L8   quantize + uint8 level indices     x -> q + levels (C ABI)            9 B/element
    K1   quantize alone, for reference                                         8 B/element
    K3   inv_scale_down                     u -> y                             8 B/element
    K6   point gradient                     g + uint8 indices -> k sums        5 B/element
Usage: python tools/side_output_probe.py [buckets, comma separated] [k values, comma separated]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import _lib  # noqa: E402

if os.environ.get('QD_LIB'):                      # A/B on one box: another build of the library (docs/history/tools/build_rev_lib.py)
    _lib.LIB_PATH = os.environ['QD_LIB']

N = 1 << 26
dev = torch.device('cuda:0')
R = 3
buckets = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '256,100,36,33,250,7,513,1000').split(',')]
ks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '4,16').split(',')]
xs = [torch.randn(N, device=dev) for _ in range(R)]
lib = _lib.load()
ws = _lib.workspace(dev)
st = torch.cuda.current_stream().cuda_stream


def timeit(name, fn, bpe, iters=12):
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.12:
        for _ in range(10):
            fn(i)
            i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    print('%-64s %8.2f us %7.1f GB/s %5.1f%%' % (name, best, bpe * N / best / 1e3, bpe * N / best / 1e3 / 80), flush=True)


print('# library:', os.path.relpath(_lib.LIB_PATH))
for i in range(400):
    quantization.uniformQuantization(xs[i % R], 16, bucket_size=256)
torch.cuda.synchronize()
qs = [torch.empty(N, device=dev) for _ in range(R)]
levs = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(R)]
idx64 = [torch.empty(N, dtype=torch.int64, device=dev) for _ in range(2)]
for b in buckets:
    timeit('K1 quantize bucket %d' % b, lambda i, b=b: lib.qd_uniform_f32(xs[i % R].data_ptr(), qs[i % R].data_ptr(), N, b, 16, None, None,
                                                                       None, None, 0, 0.0, 0, 0, ws.data_ptr(), ws.numel(), st), 8)
    nbk = lib.qd_num_buckets(N, b)
    ab3 = torch.rand(2, nbk, device=dev)
    timeit('K3 inv_scale_down bucket %d' % b, lambda i, b=b: lib.qd_inv_scale_f32(xs[i % R].data_ptr(), qs[i % R].data_ptr(), N, b, ab3[0].data_ptr(),
                                                                             ab3[1].data_ptr(), None, st), 8)
    timeit('L8 quantize + uint8 levels bucket %d' % b,
           lambda i, b=b: lib.qd_uniform_f32(xs[i % R].data_ptr(), qs[i % R].data_ptr(), N, b, 16, None, None, levs[i % R].data_ptr(), None, 0, 0.0,
                                             0, 0, ws.data_ptr(), ws.numel(), st), 9)
    for k in ks:
        pts = torch.sort(torch.rand(k, device=dev))[0].contiguous()
        nb = lib.qd_num_buckets(N, b)
        ab = torch.empty(2, nb, device=dev)
        timeit('K4 nonUniform k=%d int64 idx bucket %d' % (k, b),
               lambda i, b=b: lib.qd_nearest_point_f32(xs[i % R].data_ptr(), 0, pts.data_ptr(), k, 0, qs[i % R].data_ptr(), idx64[i % 2].data_ptr(), 8,
                                                       N, b, ab[0].data_ptr(), ab[1].data_ptr(), None, 0, 0.0, ws.data_ptr(), ws.numel(), st), 16)
        if k <= 256:
            us = [torch.rand(N, device=dev) for _ in range(2)]
            ab.fill_(1.0)
            ab[1].zero_()
            timeit('K5 pre-processed forward k=%d uint8 idx bucket %d' % (k, b),
                   lambda i, b=b: lib.qd_nearest_point_f32(us[i % 2].data_ptr(), 1, pts.data_ptr(), k, 1, qs[i % R].data_ptr(), levs[i % R].data_ptr(), 1,
                                                           N, b, ab[0].data_ptr(), ab[1].data_ptr(), None, 0, 0.0, ws.data_ptr(), ws.numel(), st), 9)
            del us
            gs = [torch.randn(N, device=dev) for _ in range(2)]
            gp = torch.empty(k, device=dev)
            levs[0].random_(0, k)
            levs[1].random_(0, k)
            timeit('K6 point gradient k=%d uint8 idx bucket %d' % (k, b),
                   lambda i, b=b: lib.qd_point_grad_f32(gs[i % 2].data_ptr(), levs[i % 2].data_ptr(), 1, ab[0].data_ptr(), N, b, k, gp.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), st), 5)
            del gs
