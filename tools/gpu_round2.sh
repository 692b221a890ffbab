#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== kbench"; timeout 600 ./build/kbench > gpurun_out/kbench.log 2>&1; cat gpurun_out/kbench.log
echo "== bench"; timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json
echo "== rocprof stats"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "rc=$?"
find gpurun_out/prof_stats -type f | head; f=$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_$c.err); echo "rc=$?"
  find gpurun_out/pmc_$c -type f | head -5
  f=$(find gpurun_out/pmc_$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && (head -3 "$f"; grep k_bucket_vec "$f" | head -3)
done
