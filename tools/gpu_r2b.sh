#!/bin/bash
# fused single-bucket tuning: grid limit sweep + rocprof kernel durations
set +e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in 128 256 512 1024; do echo "== QD_FUSED_BLOCKS=$L"; QD_FUSED_BLOCKS=$L timeout 300 python tools/k1g_probe.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/k1g_limits.txt
for L in 256 1024; do
rm -rf gpurun_out/prof_k1g_$L
(cd /tmp && QD_FUSED_BLOCKS=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_k1g_$L -o k1g -- python $R/tools/k1g_probe.py > /dev/null 2> $R/gpurun_out/prof_k1g_$L.err)
f=$(find gpurun_out/prof_k1g_$L -name '*kernel_stats.csv' | head -1); echo "== kernel stats limit $L"; [ -n "$f" ] && cut -d, -f1-4,6-8 "$f" | head -30
done
