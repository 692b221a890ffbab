#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
( for U in 2 8; do QD_HIST_WSU=$U TUNE_HIST_K=16,256 timeout 300 python tools/tune_r2.py hist; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/m_tune.txt
(cd /tmp && TUNE_HIST_K=16,256 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/m_hist_prof -o hist -- python $R/tools/tune_r2.py hist > /dev/null 2> $R/gpurun_out/m_hist_prof.err)
f=$(find gpurun_out/m_hist_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python -c "
import csv,sys
for r in list(csv.reader(open('$f')))[:5]: print(r[0][:70], r[1:7])"
