#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== correctness"; timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_property.py -x -q -m gpu -k "not 4_5_billion" 2>&1 | tail -2
echo "== correctness, chunk kernels for every size"; QD_WAVE_ANY=0 timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_per_bucket or many_chunks or extreme or stochastic" 2>&1 | tail -2
( TUNE_BUCKETS=33,50,250,7,3,5,129,255 timeout 300 python tools/tune_r2.py chunk
  QD_WAVE_ANY=0 TUNE_BUCKETS=511,513,1001 timeout 300 python tools/tune_r2.py chunk ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t_tune.txt
