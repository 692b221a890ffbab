#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for b in 1 2 4 8; do QD_HIST_BPC=$b python tools/tune_r2.py hist 2>&1 | grep -v amdgpu.ids | grep -v "k=64\|k=256"; done
} | tee gpurun_out/tune_hist.txt
