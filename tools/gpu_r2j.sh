#!/bin/bash
# Round 2, session 2, call 1: the native common-case path (correctness + host cost) and the A/B variants of the
# histogram / odd-bucket kernels (env switches, read once per process).
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== new tests"; timeout 900 python -m pytest tests/test_hip_common_path.py tests/test_hip_parity.py -x -q -m gpu -k "common or slab or general_path or lazy_arg or shapes_and or uniform_golden or random_sweep or level_histogram or many_chunks or extreme_scales" > gpurun_out/j_pytest.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/j_pytest.log
for E in "QD_HIST_PF=1 QD_HIST_U=2" "QD_HIST_PF=1 QD_HIST_U=4" "QD_HIST_REG=0 QD_HIST_ATOMIC=2"; do
  echo "== hist correctness $E"; env $E timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "level_histogram" 2>&1 | tail -2
done
echo "== chunk_any PF correctness"; QD_CHUNK_PF=8 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "many_chunks or extreme_scales or stochastic" 2>&1 | tail -2
echo "== api overhead"; timeout 600 python tools/profile_api_overhead.py > gpurun_out/j_api.txt 2>&1; head -12 gpurun_out/j_api.txt; grep -A9 "per-parameter" gpurun_out/j_api.txt
echo "== tune"
( timeout 300 python tools/tune_r2.py hist
  for E in "QD_HIST_PF=1 QD_HIST_U=2" "QD_HIST_PF=1 QD_HIST_U=4"; do env $E TUNE_HIST_K=4,16 timeout 300 python tools/tune_r2.py hist; done
  for E in "QD_HIST_REG=0 QD_HIST_ATOMIC=2" "QD_HIST_REG=0 QD_HIST_ATOMIC=4"; do env $E TUNE_HIST_K=16,64,256 timeout 300 python tools/tune_r2.py hist; done
  TUNE_BUCKETS=33,50,250,7,511 timeout 300 python tools/tune_r2.py chunk
  for P in 6 8; do QD_CHUNK_PF=$P TUNE_BUCKETS=33,50,250,7,511 timeout 300 python tools/tune_r2.py chunk; done
) 2>&1 | grep -v "amdgpu.ids" > gpurun_out/j_tune.txt
cat gpurun_out/j_tune.txt
