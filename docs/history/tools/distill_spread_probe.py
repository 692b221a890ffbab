#!/usr/bin/env python3
"""Where does the repetition-to-repetition spread of the configs[1] distillation step come from?

    python tools/distill_spread_probe.py                       # wall clock + per-step GPU time (HIP events) per repetition
    rocprofv3 --kernel-trace --output-format csv -d DIR -o spread -- python tools/distill_spread_probe.py --marks
    python tools/distill_spread_probe.py --analyse DIR          # splits the kernel trace at the marker kernels: per
                                                                # repetition GPU-busy time, idle time between kernels,
                                                                # and which kernels differ between the fastest and the
                                                                # slowest repetition

BENCH_r02 listed 100-step repetitions of the same trainer at 421.6 / 417.6 / 519.1 steps/s.  The step is ~2 ms of ~170
small kernels, so it is either bound by the GPU executing them (then the kernels' own durations differ) or by the host
issuing them (then the gaps differ).  --marks launches a fill kernel with a unique grid size between repetitions; the
analysis cuts the trace there.
"""
import argparse
import collections
import csv
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MARK_BASE = 7_340_033            # elements of the marker fill of repetition r: MARK_BASE + 4096 * r


def run(args):
    import torch
    from harness import models
    from harness.distill import DistillTrainer, synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    tr = DistillTrainer(models.student(), models.teacher(), dev, num_bits=4, bucket_size=256, mode=args.mode)
    batches = [synthetic_batch(50, dev, seed=i) for i in range(4)]
    for i in range(60):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    marks = [torch.empty(MARK_BASE + 4096 * r, device=dev) for r in range(args.reps + 1)] if args.marks else None
    rows = []
    for r in range(args.reps):
        if marks:
            marks[r].fill_(1.0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        host_issue = 0.0
        for i in range(args.steps):
            a = time.perf_counter()
            tr.step(*batches[i % 4])
            host_issue += time.perf_counter() - a
            evs[i + 1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        rows.append({'rep': r, 'wall_ms_per_step': wall / args.steps * 1e3, 'steps_per_sec': args.steps / wall,
                     'host_issue_ms_per_step': host_issue / args.steps * 1e3,
                     'gpu_step_ms_median': statistics.median(per_step), 'gpu_step_ms_p10': sorted(per_step)[len(per_step) // 10],
                     'gpu_step_ms_p90': sorted(per_step)[9 * len(per_step) // 10], 'gpu_step_ms_max': max(per_step)})
        if args.sleep and r % 3 == 2:
            time.sleep(args.sleep)                  # an idle gap: does the next repetition start slower (clock ramp)?
    if marks:
        marks[args.reps].fill_(1.0)
        torch.cuda.synchronize()
    print('mode %s, %d repetitions of %d steps, batch 50' % (args.mode, args.reps, args.steps))
    print('%3s %10s %10s %12s %12s %10s %10s %10s' % ('rep', 'steps/s', 'wall ms', 'host-issue ms', 'GPU med ms', 'p10', 'p90', 'max'))
    for w in rows:
        print('%3d %10.1f %10.4f %12.4f %12.4f %10.4f %10.4f %10.4f' % (w['rep'], w['steps_per_sec'], w['wall_ms_per_step'],
                                                                       w['host_issue_ms_per_step'], w['gpu_step_ms_median'],
                                                                       w['gpu_step_ms_p10'], w['gpu_step_ms_p90'], w['gpu_step_ms_max']))
    sps = [w['steps_per_sec'] for w in rows]
    print('steps/s: min %.1f  median %.1f  max %.1f  (max/min = %.3f)' % (min(sps), statistics.median(sps), max(sps), max(sps) / min(sps)))
    print(json.dumps({'rows': rows}))


def analyse(d):
    files = [f for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)]
    if not files:
        print('no *kernel_trace.csv under', d)
        return 1
    recs = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                try:
                    recs.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), row['Kernel_Name'],
                                 int(row.get('Grid_Size_X') or row.get('Grid_Size') or 0), int(row.get('Workgroup_Size_X') or 1)))
                except (KeyError, ValueError):
                    continue
    recs.sort()
    # markers: the fill kernels whose grid covers MARK_BASE + 4096 r elements (any vector width 1 / 2 / 4 / 8)
    def mark_index(name, grid):
        if 'fill' not in name.lower() and 'Fill' not in name:
            return None
        for vec in (1, 2, 4, 8, 16):
            for r in range(0, 64):
                n = MARK_BASE + 4096 * r
                blocks_elems = -(-n // vec)
                if abs(grid - blocks_elems) <= 1024 * 8 and grid >= blocks_elems - 1:
                    return r
        return None
    cuts = []
    for i, (s, e, name, grid, wg) in enumerate(recs):
        r = mark_index(name, grid)
        if r is not None and grid > 500000:
            cuts.append((i, r))
    if len(cuts) < 3:
        print('markers not found (%d); kernels with the largest grids:' % len(cuts))
        for s, e, name, grid, wg in sorted(recs, key=lambda t: -t[3])[:8]:
            print('  ', grid, wg, name[:100])
        return 1
    reps = []
    for (i0, r0), (i1, r1) in zip(cuts, cuts[1:]):
        seg = recs[i0 + 1:i1]
        if not seg:
            continue
        busy = sum(e - s for s, e, *_ in seg)
        span = seg[-1][1] - seg[0][0]
        # idle = span minus the union of the kernel intervals
        union, cur_s, cur_e = 0, seg[0][0], seg[0][1]
        for s, e, *_ in seg[1:]:
            if s > cur_e:
                union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        union += cur_e - cur_s
        per_kernel = collections.defaultdict(lambda: [0, 0])
        for s, e, name, *_ in seg:
            per_kernel[name][0] += e - s
            per_kernel[name][1] += 1
        reps.append({'rep': len(reps), 'kernels': len(seg), 'span_ms': span / 1e6, 'busy_ms': busy / 1e6, 'covered_ms': union / 1e6,
                     'idle_ms': (span - union) / 1e6, 'per_kernel': per_kernel})
    print('%3s %8s %10s %10s %10s %10s' % ('rep', 'kernels', 'span ms', 'sum-of-kernels', 'covered ms', 'idle ms'))
    for w in reps:
        print('%3d %8d %10.2f %10.2f %10.2f %10.2f' % (w['rep'], w['kernels'], w['span_ms'], w['busy_ms'], w['covered_ms'], w['idle_ms']))
    fast = min(reps, key=lambda w: w['span_ms'])
    slow = max(reps, key=lambda w: w['span_ms'])
    print('fastest repetition %d: %.2f ms, slowest %d: %.2f ms (x %.3f); of the difference %.2f ms, kernels account for %.2f ms and '
          'idle gaps for %.2f ms' % (fast['rep'], fast['span_ms'], slow['rep'], slow['span_ms'], slow['span_ms'] / fast['span_ms'],
                                     slow['span_ms'] - fast['span_ms'], slow['covered_ms'] - fast['covered_ms'],
                                     slow['idle_ms'] - fast['idle_ms']))
    names = sorted(slow['per_kernel'], key=lambda k: -(slow['per_kernel'][k][0] - fast['per_kernel'].get(k, [0, 0])[0]))
    print('kernels by (slow - fast) total time:')
    print('%10s %10s %8s  %s' % ('fast ms', 'slow ms', 'calls', 'kernel'))
    for k in names[:12]:
        print('%10.3f %10.3f %8d  %s' % (fast['per_kernel'].get(k, [0, 0])[0] / 1e6, slow['per_kernel'][k][0] / 1e6, slow['per_kernel'][k][1], k[:110]))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=12)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--mode', default='multi')
    ap.add_argument('--marks', action='store_true')
    ap.add_argument('--sleep', type=float, default=0.0)
    ap.add_argument('--analyse', default=None)
    a = ap.parse_args()
    if a.analyse:
        return analyse(a.analyse)
    run(a)
    return 0


if __name__ == '__main__':
    sys.exit(main())
