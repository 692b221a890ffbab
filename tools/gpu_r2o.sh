#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
( for S in 1 0; do QD_WAVE_ANY=$S TUNE_BUCKETS=992,1000,1008,1016,1056,1024,1504,1536,1500 timeout 300 python tools/tune_r2.py chunk; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/o_tune.txt
