#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, rocprof (stats, then PMC passes), tuning tables.
# Outputs -> gpurun_out/ ; summarise afterwards with tools/summarize_profiles.py rNN
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json
echo "== bench short (driver-like K/W)"; timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-distill 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
if [ -x build/kbench ]; then echo "== kbench"; timeout 300 ./build/kbench > gpurun_out/kbench.log 2>&1; timeout 300 ./build/kbench sustained bits > gpurun_out/kbench_sustained.log 2>&1; head -20 gpurun_out/kbench.log; fi
echo "== sustain probe"; timeout 300 python tools/sustain_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sustain_probe.log; head -8 gpurun_out/sustain_probe.log
echo "== rocprof stats"
rm -rf gpurun_out/prof_stats gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-distill > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "rc=$?"
f=$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 "$f"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o bench -- python $R/bench.py --steps 10 --warmup 2 --precondition-s 0.05 --no-cpu-baseline --no-distill > /dev/null 2> $R/gpurun_out/pmc_$c.err); echo "pmc $c rc=$?"
done
