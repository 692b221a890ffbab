#!/usr/bin/env python3
"""Achieved algorithmic bandwidth of every kernel on the path (SURVEY.md 8d "other kernels"),
through the public API / C ABI, steady state, 4 rotating buffer sets.  Prints a table and writes
gpurun_out/kernels.json.   python tools/bench_kernels.py [N_log2=26]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import _lib  # noqa: E402
from quantized_distillation_amd.multi_tensor import MultiTensorQuantizer  # noqa: E402

LOG2 = int(sys.argv[1]) if len(sys.argv) > 1 else 26
N = 1 << LOG2
dev = torch.device('cuda:0')
lib = _lib.load()
R = 4
xs = [torch.randn(N, device=dev) for _ in range(R)]
gs = [torch.randn(N, device=dev) for _ in range(R)]
outs = [torch.empty(N, device=dev) for _ in range(R)]
rows = []


def timeit(name, fn, bytes_per_elem, iters=40, n=N, note=''):
    import time as _t
    t0 = _t.perf_counter()
    i = 0
    while True:                                # >= 100 ms of warm-up: past the idle-to-busy clock transient and past
        for _ in range(30):                    # the slow first tens of milliseconds on freshly allocated memory
            fn(i)
            i += 1
        torch.cuda.synchronize()
        if _t.perf_counter() - t0 > 0.1:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    gbps = bytes_per_elem * n / best / 1e3
    rows.append(dict(kernel=name, us=round(best, 2), algorithmic_B_per_elem=bytes_per_elem, GBps=round(gbps, 1),
                     frac_of_8TBps=round(gbps / 8000, 4), note=note))
    print('%-46s %9.2f us  %5.1f B/elem  %8.1f GB/s  %5.1f%% of 8 TB/s  %s' % (name, best, bytes_per_elem, gbps,
                                                                           gbps / 80, note))


# every input set is allocated before the preconditioning: kernels run 5-10 % slower for tens of
# milliseconds on freshly hipMalloc'ed memory (tools/placement_probe.py), which is neither the data nor the
# placement -- the same buffers measure 87 us later
xr = [torch.randn(N + 17, device=dev) for _ in range(R)]
xw = [torch.randn(N, device=dev).mul_(0.05) for _ in range(R)]

# precondition the chip (and touch every buffer)
for i in range(1500):
    quantization.uniformQuantization((xs, xr, xw)[(i // R) % 3][i % R], 16, bucket_size=256)
torch.cuda.synchronize()

live = [None] * R


def k1(i):
    live[i % R] = quantization.uniformQuantization(xs[i % R], 16, bucket_size=256)[0]


def k1_2bit(i):
    live[i % R] = quantization.uniformQuantization(xs[i % R], 4, bucket_size=256)[0]


def k1g(i):
    live[i % R] = quantization.uniformQuantization(xs[i % R], 16)[0]


def k1s(i):
    live[i % R] = quantization.uniformQuantization(xs[i % R], 16, bucket_size=256, stochastic_rounding=True)[0]


timeit('K1  uniform 4-bit bucket 256 (API)', k1, 8)
timeit('K1  uniform 2-bit bucket 256 (API)', k1_2bit, 8)
if LOG2 == 26:                                     # SURVEY 8d also asks for the decimal size
    xd = [x[:64000000] for x in xs]
    timeit('K1  uniform 4-bit bucket 256, N = 64,000,000', lambda i: live.__setitem__(i % R, quantization.uniformQuantization(xd[i % R], 16, bucket_size=256)[0]), 8, n=64000000)
    del xd
timeit('K1  uniform 4-bit bucket 256, ragged N = 64Mi+17', lambda i: live.__setitem__(i % R, quantization.uniformQuantization(xr[i % R], 16, bucket_size=256)[0]), 8, n=N + 17)
del xr
timeit('K1  uniform 4-bit bucket 256, weight-like 0.05*randn', lambda i: live.__setitem__(i % R, quantization.uniformQuantization(xw[i % R], 16, bucket_size=256)[0]), 8)
del xw
timeit('K1g uniform 4-bit no buckets (API, 3 launches)', k1g, 12)
timeit('K1s uniform 4-bit bucket 256 stochastic', k1s, 8)
for b in (64, 128, 512, 1024, 2048, 4096, 8192, 100, 36, 33, 50, 250, 513, 1000, 1001, 2000, 3000, 5000, 8000):
    timeit('K1  uniform 4-bit bucket %d' % b, lambda i, b=b: live.__setitem__(i % R, quantization.uniformQuantization(xs[i % R], 16, bucket_size=b)[0]), 8,
           iters=40 if b in (64, 128, 512, 1024, 2048) else 12)

sf = quantization.ScalingFunction('linear', False, False, 256)
us = [None] * R


def k2(i):
    us[i % R] = sf.scale_down(xs[i % R])


timeit('K2  scale_down bucket 256', k2, 8)
u = sf.scale_down(xs[0])
timeit('K3  inv_scale_down bucket 256', lambda i: live.__setitem__(i % R, sf.inv_scale_down(u)), 8)

for k in (4, 16, 256):
    pts = torch.sort(torch.rand(k, device=dev))[0]
    timeit('K4  nonUniform k=%d bucket 256 (int64 idx)' % k,
           lambda i, pts=pts: live.__setitem__(i % R, quantization.nonUniformQuantization(xs[i % R], pts, bucket_size=256)[0]), 16)
    fns = [quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=xs[j]) for j in range(2)]
    timeit('K5  diff-quant forward k=%d (u resident, u8 idx)' % k, lambda i, pts=pts: fns[i % 2].forward(None, pts), 9 if k <= 256 else 16)
    fns[0].forward(None, pts)
    fns[1].forward(None, pts)
    timeit('K6  point gradient k=%d (u8 idx)' % k, lambda i: fns[i % 2].backward(gs[i % R]), 5)
    del fns

fq = [quantization.uniformQuantization_variable(16, bucket_size=256) for _ in range(2)]
for j in range(2):
    fq[j].saved_for_backward = {'input': xs[j]}


def k7(i):
    f = fq[i % 2]
    f.saved_for_backward = {'input': xs[i % 2]}
    live[i % R] = f.backward(gs[i % R])


timeit("K7  'complicated' STE backward bucket 256", k7, 12)
timeit("K8  truncated STE grad mask, 32% of |w| > 1 (adversarial)", lambda i: lib.qd_truncated_ste_f32(xs[i % R].data_ptr(), gs[i % R].data_ptr(), N, 1.0, _lib.stream_ptr()), 12,
       note='w read + g read-modify-write on 79% of the float4s')
ws_ = [x * 0.2 for x in xs]
timeit("K8  truncated STE grad mask, clamped weights (nothing masked)", lambda i: lib.qd_truncated_ste_f32(ws_[i % R].data_ptr(), gs[i % R].data_ptr(), N, 1.0, _lib.stream_ptr()), 4,
       note='w read only')
timeit("K8  clamp weights to [-1, 1] (nothing out of range)", lambda i: lib.qd_clamp_f32(ws_[i % R].data_ptr(), N, 1.0, _lib.stream_ptr()), 4, note='w read only')
del ws_

from quantized_distillation_amd import codec  # noqa: E402
pks = [None] * R
timeit('PK  pack 4-bit levels + alpha/beta, bucket 256', lambda i: pks.__setitem__(i % R, codec.pack_uniform(xs[i % R], 16, 256)), 4.5,
       note='4 B read + 0.5 B written')
pk0 = codec.pack_uniform(xs[0], 16, 256)
timeit('UPK unpack 4-bit -> fp32, bucket 256', lambda i: live.__setitem__(i % R, pk0.unpack()), 4.5, note='0.5 B read + 4 B written')
lev8 = [torch.randint(0, 16, (N,), dtype=torch.uint8, device=dev) for _ in range(R)]
timeit('HST histogram of uint8 levels, k=16', lambda i: codec.histogram_u8(lev8[i % R], 16), 1)
timeit('HST histogram of uint8 levels, k=256', lambda i: codec.histogram_u8(lev8[i % R], 256), 1)
del lev8, pks, pk0

# multi-tensor over a Wide_ResNet-16-22-like set of shapes (60 tensors, 82.7 M params)
shapes = [(16, 3, 3, 3), (16,)]
w = [16, 352, 704, 1408]
for a, b in zip(w[:-1], w[1:]):
    for blk in range(2):
        cin = a if blk == 0 else b
        shapes += [(cin,), (cin,), (b, cin, 3, 3), (b,), (b,), (b,), (b, b, 3, 3), (b,)]
        if blk == 0:
            shapes += [(b, cin, 1, 1), (b,)]
shapes += [(1408,), (1408,), (10, 1408), (10,)]
masters = [torch.randn(*s, device=dev) for s in shapes]
tot = sum(m.numel() for m in masters)
mt = MultiTensorQuantizer(masters, 16, 256)
timeit('K9  multi-tensor, %d tensors %.1f M params' % (len(masters), tot / 1e6), lambda i: mt.quantize(check_pointers=False), 8, n=tot)
timeit('    same tensors, per-tensor API loop', lambda i: [quantization.uniformQuantization(m, 16, bucket_size=256) for m in masters], 8, n=tot, iters=10)
from quantized_distillation_amd.multi_tensor import MultiTensorDiffQuant  # noqa: E402
qs = [torch.empty_like(m) for m in masters]
gr = [torch.randn_like(m) for m in masters]
mdq = MultiTensorDiffQuant(masters, qs, gr, 4, 256)
ptsm = torch.sort(torch.rand(len(masters), 4, device=dev), dim=1)[0].contiguous()
timeit('K5m multi-tensor assign, %d tensors %.1f M, k=4' % (len(masters), tot / 1e6), lambda i: mdq.forward(ptsm), 9, n=tot)
timeit('K6m multi-tensor point gradient, k=4', lambda i: mdq.backward(), 5, n=tot)
del mdq, qs, gr
from harness import models  # noqa: E402
st = [p.data.to(dev) for p in models.student().parameters()]
tot = sum(m.numel() for m in st)
mt2 = MultiTensorQuantizer(st, 16, 256)
timeit('K9  multi-tensor, CIFAR student 22 tensors 1.0 M', lambda i: mt2.quantize(check_pointers=False), 8, n=tot, iters=200)
timeit('    same, per-tensor API loop', lambda i: [quantization.uniformQuantization(m, 16, bucket_size=256) for m in st], 8, n=tot, iters=50)

os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/kernels.json', 'w') as f:
    json.dump(dict(n=N, rows=rows, device=torch.cuda.get_device_name(0)), f, indent=1)
