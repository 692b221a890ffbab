"""A/B two builds of libqd_hip.so on ONE box (boxes of the pool differ by 3 % on K1 and up to 20 % on the
write-heavy K4, so before/after numbers from different gpurun calls are not comparable):
    hipcc <flags of quantized_distillation_amd/build.py> <old sources> -o build/libqd_hip_old.so
    QD_LIB=$PWD/build/libqd_hip_old.so python tools/ab_lib.py ; python tools/ab_lib.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_distillation_amd import _lib
if os.environ.get('QD_LIB'):
    _lib.LIB_PATH = os.environ['QD_LIB']
import torch, quantization
N = 64 << 20
R = 4
xs = [torch.randn(N, device='cuda') for _ in range(R)]
live = [None] * R
def timeit(name, fn, iters=30):
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.15:
        for _ in range(20): fn(i); i += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for i in range(iters): fn(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    print('%-10s %-40s %8.2f us' % (os.path.basename(_lib.LIB_PATH)[:10], name, best), flush=True)
timeit('K1 b256', lambda i: live.__setitem__(i % R, quantization.uniformQuantization(xs[i % R], 16, bucket_size=256)[0]))
for k in (4, 256):
    pts = torch.sort(torch.rand(k, device='cuda'))[0]
    timeit('K4 k=%d' % k, lambda i: live.__setitem__(i % R, quantization.nonUniformQuantization(xs[i % R], pts, bucket_size=256)[0]))
    fns = [quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=xs[j]) for j in range(2)]
    timeit('K5 k=%d' % k, lambda i: fns[i % 2].forward(None, pts))
    del fns
