#!/usr/bin/env python3
"""Achieved algorithmic bandwidth of every kernel on the path, through the public API / C ABI: the rows of
harness/kernel_bench.py (the same code bench.py's `kernels` leg runs) plus the bucket-size / point-count sweeps.  Prints
the table and writes gpurun_out/kernels.json.

    python tools/bench_kernels.py [N_log2=26] [--no-sweeps] [--only=HUF,LVH,...]     (--only: rows whose name starts with one of these tags)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from harness import kernel_bench  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
log2n = int(args[0]) if args else 26
dev = torch.device('cuda:0')
print('# %s; method: %s' % (torch.cuda.get_device_name(0), kernel_bench.__doc__.split('Method, per row:')[1].split('\n\n')[0].replace('\n', ' ')))
only = [a.split('=', 1)[1].split(',') for a in sys.argv[1:] if a.startswith('--only=')]
rows = kernel_bench.run(dev, log2n=log2n, sweeps='--no-sweeps' not in sys.argv, verbose=True,
                        only=(lambda name: any(name.startswith(t) for t in only[0])) if only else None)
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/kernels.json', 'w') as f:
    json.dump(dict(n=1 << log2n, rows=rows, device=torch.cuda.get_device_name(0)), f, indent=1)
