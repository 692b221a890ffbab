#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== wave_any correctness (align 32)"; timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_per_bucket or many_chunks or extreme_scales" 2>&1 | tail -2
echo "== align 16, all sizes"; QD_WAVE_ALIGN=16 QD_WAVE_ANY=2 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_per_bucket" 2>&1 | tail -2
( for A in 4 16 32; do QD_WAVE_ALIGN=$A QD_WAVE_ANY=2 TUNE_BUCKETS=300,511,513,1000,1001,1016,1500,1536,2000,3000,5000,8000 timeout 300 python tools/tune_r2.py chunk; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/p_tune.txt
