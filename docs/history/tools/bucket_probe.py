"""Timing of uniformQuantization for bucket sizes off the vector path (64 Mi fp32, 16 levels).
Usage (GPU box): python tools/bucket_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import quantization

dev = 'cuda:0'
N = 64 << 20
R = 3
xs = [torch.randn(N, device=dev) for _ in range(R)]
live = [None] * R
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for i in range(50):
        quantization.uniformQuantization(xs[i % R], 16, bucket_size=256)
    torch.cuda.synchronize()
for b in [int(v) for v in os.environ.get("QD_PROBE_BUCKETS", "256,100,1000,36,300,2000,4096,8192,12,4,20,52,33,7").split(",")]:
    def fn(i):
        live[i % R] = quantization.uniformQuantization(xs[i % R], 16, bucket_size=b)[0]
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(10):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print('bucket %5d : %8.2f us  %7.1f GB/s' % (b, best, 8 * N / best / 1e3), flush=True)
