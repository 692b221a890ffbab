#!/usr/bin/env python3
"""Authoring-container only (needs /root/reference): time the REFERENCE's own uniformQuantization on the headline
workload next to the two CPU ports that bench.py times on the GPU box (oracle/qd_oracle.c, oracle/torch_port.py), on
the same cores.  The GPU box has no /root/reference, so bench.py's cpu_baseline is kind "port"; this file calibrates
the ports against the real thing on identical hardware.  Writes docs/history/profiles/r01_reference_cpu_timing.json."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
N, S, BUCKET = 64 << 20, 16, 256
threads = os.cpu_count()
torch.set_num_threads(threads)
x = torch.randn(N, generator=torch.Generator().manual_seed(0))


def timed(fn, runs=5):
    fn()                                       # warm-up
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return dict(min_s=round(min(ts), 4), median_s=round(sorted(ts)[len(ts) // 2], 4),
                GBps_at_min=round(8 * N / min(ts) / 1e9, 2))


out = dict(workload='uniformQuantization(randn(64Mi, seed 0), s=16, bucket_size=256), algorithmic 8 B/element',
           cpu_count=threads, cpu_model=[l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0],
           torch=torch.__version__)

# 1. the reference itself
sys.path.insert(0, REF)
import quantization as refq                    # noqa: E402
assert os.path.abspath(refq.__file__).startswith(REF)
out['reference'] = timed(lambda: refq.uniformQuantization(x, S, bucket_size=BUCKET))
q_ref = refq.uniformQuantization(x, S, bucket_size=BUCKET)[0].numpy()
sys.path.remove(REF)
for m in [k for k in sys.modules if k == 'quantization' or k.startswith('quantization.')]:
    del sys.modules[m]

# 2. the ports bench.py times on the GPU box
sys.path.insert(0, ROOT)
from oracle import oracle_c                    # noqa: E402
from oracle.torch_port import uniform_quantize_torch_ops   # noqa: E402
oracle_c.build()
xn = x.numpy()
out['c_port_openmp'] = dict(threads=oracle_c.max_threads(), **timed(lambda: oracle_c.uniform_quantize(xn, S, BUCKET, want_idx=False, want_lev=False)))
out['torch_op_port'] = dict(threads=threads, **timed(lambda: uniform_quantize_torch_ops(x, S, BUCKET)))
out['ports_bit_identical_to_reference'] = bool(
    np.array_equal(oracle_c.uniform_quantize(xn, S, BUCKET, want_idx=False, want_lev=False)['q'], q_ref) and
    np.array_equal(uniform_quantize_torch_ops(x, S, BUCKET)[0].numpy(), q_ref))
with open(os.path.join(ROOT, 'profiles', 'r01_reference_cpu_timing.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
