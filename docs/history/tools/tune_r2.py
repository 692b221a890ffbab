#!/usr/bin/env python3
"""Round-2 tuning table: the kernels VERDICT r01 #5 names (K6 with many points, the level histogram, odd bucket
sizes), N = 64 Mi, steady state.  Environment switches of the library (read once per process) select the variants:
QD_PG_U=8|16|32 (K6 loads in flight per lane for tables above 64 KiB), QD_WAVE_ANY=0|1|2, QD_WAVE_ALIGN=4|16|32, QD_NO_VEC=1
(the one-wave-per-bucket kernel: off / default sizes / every size above 256; window alignment in elements; vector sizes through it).
TUNE_BUCKETS / TUNE_HIST_K select the rows.  (The histogram switches of the earlier sessions -- QD_HIST_REG, QD_HIST_BPC -- went with
the kernels they selected; tools/gpu_r2c.sh / gpu_r2d.sh are kept as the record of how docs/history/profiles/r02_tune_kernels.txt's first half was made.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import codec  # noqa: E402

what = set(sys.argv[1:]) or {'k6', 'hist', 'chunk'}
N = 1 << 26
dev = torch.device('cuda:0')
R = 4
xs = [torch.randn(N, device=dev) for _ in range(R)]
live = [None] * R


def timeit(name, fn, bpe, iters=30):
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(20):
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
    print('%-58s %8.2f us %7.1f GB/s %5.1f%%' % (name, best, bpe * N / best / 1e3, bpe * N / best / 1e3 / 80), flush=True)


env = ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('QD_'))
print('# env:', env or '(defaults)')
for i in range(600):
    quantization.uniformQuantization(xs[i % R], 16, bucket_size=256)
torch.cuda.synchronize()
if 'chunk' in what:
    for b in [int(v) for v in os.environ.get('TUNE_BUCKETS', '33,50,250,7,511,513,1001,100,36,1000,12,2000').split(',')]:
        timeit('K1 uniform 4-bit bucket %d' % b,
               lambda i, b=b: live.__setitem__(i % R, quantization.uniformQuantization(xs[i % R], 16, bucket_size=b)[0]), 8, iters=12)
if 'k6' in what:
    gs = [torch.randn(N, device=dev) for _ in range(R)]
    for k in (16, 32, 64, 128, 256, 512):
        pts = torch.sort(torch.rand(k, device=dev))[0]
        fns = [quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=xs[j]) for j in range(2)]
        fns[0].forward(None, pts)
        fns[1].forward(None, pts)
        timeit('K6 point gradient k=%d (u8 idx)' % k if k <= 256 else 'K6 point gradient k=%d (int64 idx)' % k,
               lambda i: fns[i % 2].backward(gs[i % R]), 5 if k <= 256 else 12)
        del fns
    del gs
if 'hist' in what:
    for k in [int(v) for v in os.environ.get('TUNE_HIST_K', '4,16,64,256').split(',')]:
        lev8 = [torch.randint(0, k, (N,), dtype=torch.uint8, device=dev) for _ in range(R)]
        timeit('HST histogram of uint8 levels, k=%d' % k, lambda i: codec.histogram_u8(lev8[i % R], k), 1)
        # levels as the quantizer produces them (bell-shaped: most symbols in a few bins)
        q = (torch.randn(N, device=dev) * (k / 6.0) + k / 2.0).round_().clamp_(0, k - 1).to(torch.uint8)
        timeit('HST histogram, k=%d, bell-shaped symbols' % k, lambda i: codec.histogram_u8(q, k), 1)
        del lev8, q
