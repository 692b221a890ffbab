"""Is the 0.05*randn row of bench_kernels slower because of the data or because of where the
tensors landed?  Groups of 3 rotating inputs: A randn; B 0.05*randn (fresh allocations made the way
bench_kernels makes them); D = B's values copied into A's buffers; E = A's values copied into B's buffers.
Usage (GPU box): python tools/placement_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization

dev = 'cuda:0'
N = 64 << 20
R = 3
live = [None] * R


def timeit(name, xs, iters=40):
    def fn(i):
        live[i % R] = quantization.uniformQuantization(xs[i % R], 16, bucket_size=256)[0]
    for i in range(30):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    print('%-58s min %7.2f  med %7.2f us   ptrs %s' % (name, min(ts), sorted(ts)[2], ' '.join(hex(x.data_ptr()) for x in xs)), flush=True)


A = [torch.randn(N, device=dev) for _ in range(R)]
for i in range(1500):
    quantization.uniformQuantization(A[i % R], 16, bucket_size=256)
torch.cuda.synchronize()
timeit('A  randn', A)
xr = [torch.randn(N + 17, device=dev) for _ in range(R)]
timeit('   ragged N+17', xr)
del xr
B = [0.05 * torch.randn(N, device=dev) for _ in range(R)]
timeit('B  0.05*randn (fresh allocations)', B)
timeit('A  randn again', A)
savedA = [a.clone() for a in A]
for a, b in zip(A, B):
    a.copy_(b)
timeit('D  0.05*randn values in A buffers', A)
for b, s in zip(B, savedA):
    b.copy_(s)
timeit('E  randn values in B buffers', B)
C = [torch.randn(N, device=dev) * 1.0 for _ in range(R)]
timeit('C  randn*1.0 (fresh allocations, same recipe as B)', C)
C2 = [torch.randn(N, device=dev).mul_(0.05) for _ in range(R)]
timeit('C2 randn.mul_(0.05) in place (no temporaries)', C2)
