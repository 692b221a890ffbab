#!/usr/bin/env python3
"""Joins gpurun_out/pmc_plan.json with the two rocprofv3 --pmc passes of tools/pmc_probe.py
(gpurun_out/pmcK_FETCH_SIZE, pmcK_WRITE_SIZE) into profiles/<tag>_kernels_pmc.json."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
plan = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_plan.json')))


def per_kernel(counter):
    f = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmcK_%s' % counter, '**', '*counter_collection.csv'), recursive=True))
    rows = {}
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != counter or r['Dispatch_Id'] in seen:
            continue
        seen.add(r['Dispatch_Id'])
        rows.setdefault(r['Kernel_Name'], []).append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


fetch, write = per_kernel('FETCH_SIZE'), per_kernel('WRITE_SIZE')
out = []
for p in plan:
    def pick(table):
        for name, vals in table.items():
            if p['kernel'] in name:
                sel = vals[p['first']:p['first'] + p['reps']]
                return sum(sel) / len(sel) * 1024 if sel else None
        return None
    rd, wr = pick(fetch), pick(write)
    rec = dict(label=p['label'], kernel=p['kernel'], algorithmic_bytes=p['algorithmic_bytes'])
    if rd is not None and wr is not None:
        # FETCH_SIZE counts a 128-byte request of a 16 B/lane streaming read as 64 B on gfx950: double it
        rec.update(read_bytes=2 * rd, write_bytes=wr, hbm_bytes=2 * rd + wr,
                   traffic_over_algorithmic=round((2 * rd + wr) / p['algorithmic_bytes'], 4))
    out.append(rec)
with open(os.path.join(ROOT, 'profiles', '%s_kernels_pmc.json' % tag), 'w') as f:
    json.dump(dict(n=plan[0]['n'], note='read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md HBM); '
                   'the 2x calibration holds for 16 B/lane streams; narrower index streams may be over-corrected',
                   kernels=out), f, indent=1)
for r in out:
    print(r)
