# What a failed hipGraph capture leaves behind, and the c10d-watchdog provocation (tests/capture_worker.py), in one lease:
#   gpurun -- 'bash tools/capture_probe.sh'        -> gpurun_out/long_capture_*.log, failed_*.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/$tag.log 2>&1; echo "$tag rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|Warning\|return Variable\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/$tag.log | tail -${TAILN:-4} | cut -c1-250; }
for mode in thread_local global; do
  run long_capture_$mode python tests/capture_worker.py --long-capture 0.5 --mode $mode
done
for how in raise sync item; do
  run failed_$how python tests/capture_worker.py --failed-capture $how
done
