#!/bin/bash
# per-kernel times of the radix select (tools/select_prof.py) under rocprofv3; every step bounded
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for M in ${MS:-8 32}; do
  for KIND in ${KINDS:-gauss}; do
    (cd /tmp && M=$M KIND=$KIND timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sel_$M$KIND -o sel -- python $R/tools/select_prof.py > /dev/null 2> $R/gpurun_out/sel_$M$KIND.err < /dev/null)
    f=$(find $R/gpurun_out/sel_$M$KIND -name "*kernel_stats.csv" 2>/dev/null | head -1)
    echo "M=$M KIND=$KIND"
    [ -n "$f" ] && cut -d, -f1-4 "$f" | head -10
  done
done
