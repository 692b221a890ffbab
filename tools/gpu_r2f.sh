#!/bin/bash
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4
python tools/tune_r2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tune_r2_final.txt
rm -rf gpurun_out/sq_a gpurun_out/sq_b
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/sq_a -o sq -- python $R/tools/sq_probe2.py > /dev/null 2> $R/gpurun_out/sq_a.err); echo "pass a rc=$?"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/sq_b -o sq -- python $R/tools/sq_probe2.py > /dev/null 2> $R/gpurun_out/sq_b.err); echo "pass b rc=$?"
python tools/pmc_summarize2.py gpurun_out/sq_a gpurun_out/sq_b | tee gpurun_out/sq_r2_after.txt
