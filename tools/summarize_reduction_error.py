#!/usr/bin/env python3
"""gpurun_out/reduction_error.jsonl (written by tests/errlog.py during a test run) -> a table of the error every
floating-point reduction of the path ACHIEVED, relative to the sum of the magnitudes of its terms, next to the tolerance
it is held to (north_star: 1e-6).

    python tools/summarize_reduction_error.py [log] > docs/history/profiles/r04_reduction_error.txt
"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'reduction_error.jsonl')
    rows = collections.OrderedDict()
    for line in open(path):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        d = rows.setdefault(r['kind'], {'n': 0, 'max': 0.0, 'vals': [], 'tol': r.get('tol', 1e-6), 'worst': '', 'nt': 0})
        d['n'] += 1
        d['vals'].append(r['err_over_sum_abs_terms'])
        d['nt'] = max(d['nt'], r.get('n_terms') or 0)
        if r['err_over_sum_abs_terms'] >= d['max']:
            d['max'] = r['err_over_sum_abs_terms']
            d['worst'] = r['case']
    print('error / sum|terms| achieved by every floating-point reduction check of the test run (%s)' % os.path.relpath(path, ROOT))
    print('%-92s %6s %10s %10s %8s %9s  %s' % ('check', 'checks', 'max', 'median', 'tol', 'margin', 'worst case'))
    for k, d in rows.items():
        v = sorted(d['vals'])
        med = v[len(v) // 2]
        print('%-92s %6d %10.2e %10.2e %8.0e %8.1fx  %s' % (k[:92], d['n'], d['max'], med, d['tol'],
                                                           d['tol'] / d['max'] if d['max'] > 0 else float('inf'), d['worst'][:70]))


if __name__ == '__main__':
    main()
