#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity tests"; timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_property.py -q -x 2>&1 | tail -4
{
python tools/tune_r2.py chunk hist 2>&1 | grep -v amdgpu.ids
QD_HIST_REG=0 python tools/tune_r2.py hist 2>&1 | grep -v amdgpu.ids
for u in 8 16 32; do QD_PG_U=$u python tools/tune_r2.py k6 2>&1 | grep -v amdgpu.ids; done
} | tee gpurun_out/tune_r2.txt
