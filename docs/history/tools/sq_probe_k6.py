#!/usr/bin/env python3
"""Three dispatches each of the point-gradient call (K6, uint8 indices, N = 64 Mi) for SQ-counter passes: k = 16 (LDS columns
[k][256]), 64 (the same with the merged four-element update), 128 / 256 (four waves taking turns on [k][64]), at bucket 256 and
-- for k = 16 / 128 -- at bucket 100 (the bucket walk).  `--summarize` turns gpurun_out/sq_k6 into a table."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(256, 16), (256, 64), (256, 128), (256, 256), (100, 16), (100, 128)]

if '--summarize' in sys.argv:
    d = os.path.join(ROOT, 'gpurun_out', 'sq_k6')
    cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0]))}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(cc[0])):
        name = r['Kernel_Name']
        if 'k_point_grad' not in name or 'final' in name:
            continue
        per.setdefault(int(r['Dispatch_Id']), {'name': name[name.find('k_point_grad'):].split('(')[0]})[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(per)
    for g, (b, k) in enumerate(CASES):
        grp = ids[3 * g:3 * g + 3]
        if len(grp) < 3:
            break
        keys = sorted(c for c in per[grp[0]] if c != 'name')
        avg = {c: sum(per[i].get(c, 0.0) for i in grp) / 3 for c in keys}
        us = sum(dur.get(str(i), 0.0) for i in grp) / 3
        print('K6 bucket %-4d k=%-4d %-38s %7.1f us (under the counters)  ' % (b, k, per[grp[0]]['name'], us)
              + '  '.join('%s=%.3g' % (c.replace('SQ_', ''), avg[c]) for c in keys))
    sys.exit(0)

sys.path.insert(0, ROOT)
import torch  # noqa: E402

from quantized_distillation_amd import _lib  # noqa: E402

lib = _lib.load()
N = 1 << 26
dev = torch.device('cuda:0')
gs = [torch.randn(N, device=dev) for _ in range(2)]
idx = torch.empty(N, dtype=torch.uint8, device=dev)
ws = _lib.workspace(dev)
st = torch.cuda.current_stream().cuda_stream
for b, k in CASES:
    idx.random_(0, k)
    ab = torch.ones(lib.qd_num_buckets(N, b), device=dev)
    gp = torch.empty(k, device=dev)
    torch.cuda.synchronize()
    for i in range(3):
        _lib.check(lib.qd_point_grad_f32(gs[i % 2].data_ptr(), idx.data_ptr(), 1, ab.data_ptr(), N, b, k, gp.data_ptr(), ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
print('ok')
