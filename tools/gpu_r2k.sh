#!/bin/bash
# call 2: launch-cost probe, LDS-atomic histogram A/B, kernel-trace of the histogram call
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== launch probe"; timeout 120 ./build/launch_probe | tee gpurun_out/k_launch_probe.txt
for E in "QD_HIST_REG=0 QD_HIST_ATOMIC=2"; do
  echo "== hist correctness $E"; env $E timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "level_histogram" 2>&1 | tail -2
done
( for E in "QD_HIST_REG=0 QD_HIST_ATOMIC=1" "QD_HIST_REG=0 QD_HIST_ATOMIC=2" "QD_HIST_REG=0 QD_HIST_ATOMIC=4"; do env $E TUNE_HIST_K=16,64,256 timeout 300 python tools/tune_r2.py hist; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k_tune.txt
echo "== kernel trace of the histogram call"
(cd /tmp && TUNE_HIST_K=16,256 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k_hist_prof -o hist -- python $R/tools/tune_r2.py hist > /dev/null 2> $R/gpurun_out/k_hist_prof.err)
f=$(find gpurun_out/k_hist_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -d, -f1-6 "$f" | head -8
