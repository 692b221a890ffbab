#!/usr/bin/env python3
"""Turn the rocprofv3 output that a GPU session left under gpurun_out/ into the small, tracked
summaries under profiles/ (named per round).  Usage: python tools/summarize_profiles.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
OUT = os.path.join(ROOT, 'profiles')
os.makedirs(OUT, exist_ok=True)
G = os.path.join(ROOT, 'gpurun_out')


def first(pattern):
    m = sorted(glob.glob(os.path.join(G, pattern), recursive=True))
    return m[0] if m else None


# 1. kernel stats of `rocprofv3 --kernel-trace --stats -- python bench.py ...`
st = first('prof_stats/**/*kernel_stats.csv') or first('prof_stats/*kernel_stats.csv')
if st:
    shutil.copy(st, os.path.join(OUT, '%s_bench_kernel_stats.csv' % tag))
tr = first('prof_stats/**/*kernel_trace.csv') or first('prof_stats/*kernel_trace.csv')
steady = None
if tr:
    rows = [r for r in csv.DictReader(open(tr)) if 'k_bucket_vec' in r['Kernel_Name']]
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    if d:
        tail = d[len(d) // 2:]
        steady = dict(launches=len(d), avg_us=sum(d) / len(d), min_us=min(d), max_us=max(d),
                      last_half_avg_us=sum(tail) / len(tail),
                      grid=rows[0]['Grid_Size_X'], workgroup=rows[0]['Workgroup_Size_X'],
                      vgpr=rows[0]['VGPR_Count'], sgpr=rows[0]['SGPR_Count'], lds=rows[0]['LDS_Block_Size'])
        with open(os.path.join(OUT, '%s_k_bucket_vec_durations_us.txt' % tag), 'w') as f:
            f.write('# per-dispatch duration (us) of k_bucket_vec<QDQ,16,4> in launch order, from rocprofv3 --kernel-trace\n')
            f.write(' '.join('%.1f' % v for v in d) + '\n')

# 2. PMC passes (separate runs): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
#    reports exactly half of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section)
pmc = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = first('pmc_%s/**/*counter_collection.csv' % c) or first('pmc_%s/*counter_collection.csv' % c)
    if not f:
        continue
    vals = {}
    for r in csv.DictReader(open(f)):
        if 'k_bucket_vec' in r['Kernel_Name'] and r['Counter_Name'] == c:
            vals[r['Dispatch_Id']] = float(r['Counter_Value'])
    if vals:
        v = list(vals.values())
        pmc[c] = dict(per_launch_KiB_avg=sum(v) / len(v), launches=len(v), min=min(v), max=max(v))
if pmc.get('FETCH_SIZE') and pmc.get('WRITE_SIZE'):
    fetch = pmc['FETCH_SIZE']['per_launch_KiB_avg'] * 1024
    write = pmc['WRITE_SIZE']['per_launch_KiB_avg'] * 1024
    traffic = 2.0 * fetch + write
    out = {
        'kernel': 'k_bucket_vec<MODE_QDQ,16,4>', 'workload': 'N=64Mi fp32, s=16, bucket=256',
        'FETCH_SIZE_KiB_raw': pmc['FETCH_SIZE'], 'WRITE_SIZE_KiB_raw': pmc['WRITE_SIZE'],
        'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts a 128-B request of a 16 B/lane '
                      'streaming read as 64 B; MI355X_MICROARCH.md "HBM"); write bytes = WRITE_SIZE x 1024',
        'read_bytes_per_launch': 2.0 * fetch, 'write_bytes_per_launch': write,
        'k_bucket_vec_hbm_bytes_per_launch': traffic,
        'algorithmic_bytes_per_launch': 8 * 64 * 1024 * 1024,
        'traffic_over_algorithmic': traffic / (8 * 64 * 1024 * 1024),
        'collected_with': 'rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace '
                          '--output-format csv -- python bench.py --steps 10 --warmup 2 --precondition-s 0.05 --no-cpu-baseline',
    }
    with open(os.path.join(OUT, 'pmc_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    shutil.copy(os.path.join(OUT, 'pmc_traffic.json'), os.path.join(OUT, '%s_pmc_traffic.json' % tag))

# 3. plain-text logs worth keeping
for name in ('kbench.log', 'kbench_sustained.log', 'tune.log', 'sustain_probe.log', 'bench.json', 'prof_bench.json',
             'pytest_gpu.log', 'smoke.log'):
    src = os.path.join(G, name)
    if os.path.exists(src):
        base, ext = os.path.splitext(name)
        shutil.copy(src, os.path.join(OUT, '%s_%s%s' % (tag, base, '.txt' if ext == '.log' else ext)))

print(json.dumps(dict(kernel_stats=bool(st), steady=steady, pmc=pmc), indent=1))
