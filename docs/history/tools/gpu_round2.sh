#!/bin/bash
# Round-2 closing session on one MI355X box: smoke, the whole GPU suite, bench (as the driver runs it and with the default
# flags), rocprofv3 stats + the two PMC passes of the bench command, the per-kernel table, the round-2 tuning table, the
# per-kernel PMC traffic, the API host-cost table, a property soak.  Outputs -> gpurun_out/ ; summarise with
# tools/summarize_profiles.py r02 and tools/pmc_kernels_summarize.py r02.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
echo "== bench (default flags)"; timeout 1200 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench.json
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-distill 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['extended'])"
echo "== rocprof stats"
rm -rf gpurun_out/prof_stats gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmcK_FETCH_SIZE gpurun_out/pmcK_WRITE_SIZE
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-distill > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o bench -- python $R/bench.py --steps 10 --warmup 2 --precondition-s 0.05 --no-cpu-baseline --no-distill > /dev/null 2> $R/gpurun_out/pmc_$c.err); echo "pmc $c rc=$?"
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcK_$c -o probe -- python $R/tools/pmc_probe.py > /dev/null 2> $R/gpurun_out/pmcK_$c.err); echo "pmcK $c rc=$?"
done
echo "== kernels"; timeout 900 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/kernels.txt; tail -5 gpurun_out/kernels.txt
echo "== tune"; timeout 900 python tools/tune_r2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/tune_r2.txt; tail -3 gpurun_out/tune_r2.txt
echo "== api"; timeout 600 python tools/profile_api_overhead.py 2>&1 | grep -v amdgpu.ids > gpurun_out/api_overhead.txt; head -6 gpurun_out/api_overhead.txt
echo "== soak"; QD_SOAK=10 timeout 1200 python -m pytest tests/test_hip_property.py -x -q -m gpu > gpurun_out/property_soak.log 2>&1; tail -2 gpurun_out/property_soak.log
