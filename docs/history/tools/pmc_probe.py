#!/usr/bin/env python3
"""Runs each kernel of the path a few times at N = 16 Mi so that a rocprofv3 --pmc pass can attribute
HBM traffic per kernel.  Writes gpurun_out/pmc_plan.json: for each label the kernel-name substring,
which dispatches of that kernel belong to it, and the algorithmic bytes per launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization  # noqa: E402
from quantized_distillation_amd import _lib, codec  # noqa: E402

N = 1 << 24
REPS = 3
dev = torch.device('cuda:0')
# enough distinct buffers that nothing is served from the 256 MiB Infinity Cache by accident
xs = [torch.randn(N, device=dev) for _ in range(6)]
gs = [torch.randn(N, device=dev) for _ in range(6)]
plan, counts = [], {}


def record(label, kernel, algo_bytes, fn):
    first = counts.get(kernel, 0)
    for i in range(REPS):
        fn(i)
    counts[kernel] = first + REPS
    plan.append(dict(label=label, kernel=kernel, first=first, reps=REPS, algorithmic_bytes=algo_bytes, n=N))


keep = []
record('K1 uniform 4-bit bucket 256', 'k_bucket_vec<0, 16, 4, 1>', 8 * N,
       lambda i: keep.append(quantization.uniformQuantization(xs[i], 16, bucket_size=256)[0]))
record('K1 uniform 4-bit bucket 100 (chunk kernel)', 'k_bucket_chunk<0, 8>', 8 * N,
       lambda i: keep.append(quantization.uniformQuantization(xs[i + 3], 16, bucket_size=100)[0]))
record('K1 uniform 4-bit bucket 33 (chunk_any kernel)', 'k_bucket_chunk_any<0, 8>', 8 * N,
       lambda i: keep.append(quantization.uniformQuantization(xs[i], 16, bucket_size=33)[0]))
record('K1 uniform 4-bit bucket 1000 (one wave per bucket, any size)', 'k_bucket_wave_any<0, 5, 1>', 8 * N,
       lambda i: keep.append(quantization.uniformQuantization(xs[i + 3], 16, bucket_size=1000)[0]))
record('K1 uniform 4-bit bucket 513 (one wave per bucket, any size)', 'k_bucket_wave_any<0, 3, 1>', 8 * N,
       lambda i: keep.append(quantization.uniformQuantization(xs[i], 16, bucket_size=513)[0]))
lev = [torch.randint(0, 16, (N,), dtype=torch.uint8, device=dev) for _ in range(3)]
record('HST level histogram k=16 (LDS integer atomics + fold)', 'k_hist_atomic<2>', N, lambda i: keep.append(codec.histogram_u8(lev[i], 16)))
sf = quantization.ScalingFunction('linear', False, False, 256)
record('K2 scale_down', 'k_bucket_vec<1, 16, 4, 1>', 8 * N, lambda i: keep.append(sf.scale_down(xs[i + 3])))
u = sf.scale_down(xs[0])
record('K3 inv_scale_down', 'k_inv_scale', 8 * N, lambda i: keep.append(sf.inv_scale_down(u)))
pts = torch.tensor([0.0, 0.4, 0.6, 1.0], device=dev)
record('K4 nonUniform k=4 int64 idx', 'k_bucket_vec<2, 16, 4, 1>', 16 * N,
       lambda i: keep.append(quantization.nonUniformQuantization(xs[i], pts, bucket_size=256)[:2]))
fns = [quantization.nonUniformQuantization_variable(bucket_size=256, pre_process_tensors=True, tensor=xs[i]) for i in range(3)]
# (preprocess ran K2 three more times)
counts['k_bucket_vec<1, 16, 4, 1>'] = counts.get('k_bucket_vec<1, 16, 4, 1>', 0) + 4   # + sf.scale_down(xs[0]) above
record('K5 diff-quant forward k=4 u8 idx', 'k_nearest_prescaled_stream<false>', 9 * N, lambda i: keep.append(fns[i].forward(None, pts)))
record('K6 point gradient k=4 u8 idx', 'k_point_grad_fast<4, 1, 1, 4, false, false>', 5 * N, lambda i: keep.append(fns[i].backward(gs[i])[1]))
fq = quantization.uniformQuantization_variable(16, bucket_size=256)


def k7(i):
    fq.saved_for_backward = {'input': xs[i + 3]}
    keep.append(fq.backward(gs[i + 3]))


record("K7 'complicated' STE backward", 'k_ste_backward_vec<16, 4>', 12 * N, k7)
record('PK pack 4-bit', 'k_pack_vec<16, 4, 4>', int(4.5 * N), lambda i: keep.append(codec.pack_uniform(xs[i], 16, 256)))
pk = codec.pack_uniform(xs[0], 16, 256)
counts['k_pack_vec<16, 4, 4>'] += 1
record('UPK unpack 4-bit', 'k_unpack<4>', int(4.5 * N), lambda i: keep.append(pk.unpack()))
lev8 = [torch.randint(0, 16, (N,), dtype=torch.uint8, device=dev) for _ in range(3)]
record('LVH level histogram of x in one pass', 'k_level_hist_vec<16, 4>', 4 * N,
       lambda i: keep.append(codec.level_histogram(xs[i], 16, 256)))
sfh = quantization.ScalingFunction('linear', False, False, 256)
uu = sfh.scale_down(xs[1]).view(-1)
counts['k_bucket_vec<1, 16, 4, 1>'] = counts.get('k_bucket_vec<1, 16, 4, 1>', 0) + 1
import numpy as np  # noqa: E402
import quantization.help_functions as qhf  # noqa: E402
edges = torch.from_numpy(qhf._digitize_edges(16, 1e-5)).to(dev)
record('DGH digitize + histogram of the re-scaled tensor (Huffman accounting)', 'k_hist_sym<0>', 4 * N,
       lambda i: keep.append(qhf._device_counts('digitize', uu, 16, edges)))
idx64 = torch.randint(0, 4, (N,), device=dev)
record('I64 histogram of int64 point indices', 'k_hist_sym<1>', 8 * N, lambda i: keep.append(qhf._device_counts('index', idx64, 256)))
# round 3: the calls with a side output at the bucket sizes of the chunk kernels
# (K5 -- the pre-processed forward -- is the float4 stream kernel at every bucket size: <true> when a float4 can straddle buckets)
for b, kern4, kern5 in ((100, 'k_bucket_chunk<2, 8>', 'k_nearest_prescaled_stream<false>'), (33, 'k_bucket_chunk_any<2, 8>', 'k_nearest_prescaled_stream<true>'),
                        (250, 'k_bucket_chunk_any<2, 8>', 'k_nearest_prescaled_stream<true>')):
    record('K4 nonUniform k=4 int64 idx bucket %d' % b, kern4, 16 * N,
           lambda i, b=b: keep.append(quantization.nonUniformQuantization(xs[i], pts, bucket_size=b)[:2]))
    fb = [quantization.nonUniformQuantization_variable(bucket_size=b, pre_process_tensors=True, tensor=xs[i + 3]) for i in range(3)]
    record('K5 pre-processed forward k=4 u8 idx bucket %d' % b, kern5, 9 * N, lambda i, fb=fb: keep.append(fb[i].forward(None, pts)))
    del fb
lib = _lib.load()
ws = _lib.workspace(dev)
levo = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(3)]
qo = [torch.empty(N, device=dev) for _ in range(3)]
for b, kern in ((33, 'k_bucket_chunk_any<0, 8>'), (256, 'k_bucket_vec<0, 16, 4, 1>')):
    record('L8 quantize + uint8 levels bucket %d' % b, kern, 9 * N,
           lambda i, b=b: lib.qd_uniform_f32(xs[i].data_ptr(), qo[i].data_ptr(), N, b, 16, None, None, levo[i].data_ptr(), None, 0, 0.0, 0, 0,
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
torch.cuda.synchronize()
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, 'pmc_plan.json'), 'w') as f:
    json.dump(plan, f, indent=1)
print('ok', len(plan))
