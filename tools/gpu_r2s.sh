#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== correctness"; timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_property.py -x -q -m gpu -k "not 4_5_billion" 2>&1 | tail -2
( TUNE_BUCKETS=33,50,250,7,511,100,36,12,300,400,450,500,513,1000 timeout 300 python tools/tune_r2.py chunk ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s_tune.txt
