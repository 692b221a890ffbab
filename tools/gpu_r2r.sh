#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== correctness"; QD_WAVE_ANY=2 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_per_bucket or many_chunks or extreme_scales" 2>&1 | tail -2
( QD_WAVE_ANY=2 TUNE_BUCKETS=300,511,513,1000,1001,1500,2000,3000,5000,8000 timeout 300 python tools/tune_r2.py chunk
  QD_NO_VEC=1 TUNE_BUCKETS=512,1024,4096,8192 timeout 300 python tools/tune_r2.py chunk ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r_tune.txt
