#!/bin/bash
# call 3: workspace histogram (partials + fold) vs atomics, blocks per CU
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== hist correctness (default = workspace path)"; timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_property.py -x -q -m gpu -k "histogram or huffman or pack" 2>&1 | tail -2
echo "== QD_HIST_WS=1"; QD_HIST_WS=1 timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "level_histogram" 2>&1 | tail -1
( for W in 1 2 3 4; do QD_HIST_WS=$W timeout 300 python tools/tune_r2.py hist; done
  QD_HIST_WS=0 QD_HIST_REG=0 QD_HIST_ATOMIC=1 TUNE_HIST_K=4 timeout 300 python tools/tune_r2.py hist ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l_tune.txt
