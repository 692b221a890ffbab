#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== wave_any correctness"; timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_property.py -x -q -m gpu -k "wave_per_bucket or many_chunks or extreme_scales or uniform_matches_oracle or histogram" 2>&1 | tail -6
echo "== sel=2 correctness"; QD_WAVE_ANY=2 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_per_bucket" 2>&1 | tail -2
( for S in 0 1 2; do QD_WAVE_ANY=$S TUNE_BUCKETS=300,511,513,1000,1001,1500,2000,3000,5000,8000 timeout 300 python tools/tune_r2.py chunk; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/n_tune.txt
