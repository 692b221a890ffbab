#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== soak"; QD_SOAK=25 timeout 1500 python -m pytest tests/test_hip_property.py -q 2>&1 | tail -3 | tee gpurun_out/property_soak.txt
echo "== kernels"; timeout 900 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kernels.txt
echo "== api overhead"; python tools/profile_api_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/api_overhead.txt
echo "== k1g"; python tools/k1g_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k1g_probe.txt
