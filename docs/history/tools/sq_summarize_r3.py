#!/usr/bin/env python3
"""gpurun_out/sq_cur, gpurun_out/sq_prev (rocprofv3 --pmc passes of tools/sq_probe_r3.py) -> per call (in dispatch order) the
kernel, its duration and its SQ counters, for the current library and the previous one."""
import collections
import csv
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABELS = ['K5 bucket 256 k=256', 'K5 bucket 256 k=16', 'K5 bucket 256 k=4', 'K5 bucket 100 k=4', 'K5 bucket 33 k=4', 'K5 bucket 100 k=256']
for tag in ('prev', 'cur'):
    d = os.path.join(ROOT, 'gpurun_out', 'sq_' + tag)
    cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if not cc or not kt:
        continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    per = collections.OrderedDict()
    for r in csv.DictReader(open(cc[0])):
        name = r['Kernel_Name']
        if 'k_bucket' not in name and 'k_nearest' not in name:
            continue
        per.setdefault(int(r['Dispatch_Id']), {'name': name.split('::')[-1].split('(')[0]})[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(per)
    print('== library: %s' % ('build/libqd_hip_prev.so (the round-2 kernels)' if tag == 'prev' else 'current'))
    for g, label in enumerate(LABELS):
        grp = ids[3 * g:3 * g + 3]
        if len(grp) < 3:
            break
        keys = sorted(k for k in per[grp[0]] if k != 'name')
        avg = {k: sum(per[i].get(k, 0.0) for i in grp) / 3 for k in keys}
        us = sum(dur.get(str(i), 0.0) for i in grp) / 3
        print('%-22s %-34s %8.1f us (under the counters)  ' % (label, per[grp[0]]['name'], us) + '  '.join('%s=%.3g' % (k.replace('SQ_', ''), avg[k]) for k in keys))
