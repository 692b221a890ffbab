#!/usr/bin/env python3
"""Average per-dispatch counter values per kernel from rocprofv3 --pmc CSV output directories."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            if 'anonymous namespace' not in name:
                continue
            short = name.split('::', 1)[1].split('(')[0].replace('anonymous namespace)::', '')
            acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    print('   ' + '  '.join('%s=%.0f' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
