#!/bin/bash
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/sq_a
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/sq_a -o sq -- python $R/tools/sq_probe.py > /dev/null 2> $R/gpurun_out/sq_a.err); echo "pass a rc=$?"
python tools/pmc_summarize2.py gpurun_out/sq_a | tee gpurun_out/sq_vec.txt
python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids | grep "K1 \|K9\|K1s\|K2 \|K1g"
