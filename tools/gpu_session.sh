#!/bin/bash
# One MI355X lease, driven by a list of step names:   gpurun -- 'bash tools/gpu_session.sh smoke tests driver prof'
# Every step writes under gpurun_out/ (merged back by gpurun); what is to be judged is copied to profiles/rNN_* afterwards.
#   smoke      __graft_entry__.smoke()
#   tests      the whole GPU suite (records the achieved reduction errors in gpurun_out/reduction_error.jsonl)
#   driver     the driver's exact command (python3 bench.py --gpus 1 --steps 20 --warmup 5): stdout line + bench_detail.json,
#              numbered by DRIVER_TAG; checks that the line parses from the last 6000 bytes of stdout
#   bench      bench.py with its default flags
#   prof       rocprofv3 --kernel-trace --stats of the bench command (its measuring process, `bench.py --worker`) + the two PMC passes
#   kernels    tools/bench_kernels.py (every kernel of the path, steady state);  kprof: the same under rocprofv3 --kernel-trace --stats
#   torchrun   bench.py under torch.distributed.run with one rank;  ranks2: two ranks on this one GPU through gloo
#   coverage   tools/launch_coverage.py --run: the GPU suite under rocprofv3 --kernel-trace, shipped kernels never launched
#   dispatch   tools/dispatch_map.py --trace: call geometry -> kernel map
#   div        tools/div_invariant_check.py --pairs 1e9 (the long run of the division proof)
#   soak       property tests with QD_SOAK=10
#   miopen     tools/miopen_find_probe.py: first-use search time and steps/sec per MIOPEN_FIND_MODE
#   abk8       build/ab_k8 (tools/ab_k8.hip): the truncated-STE mask variants
#   stack      ROCm / driver / torch versions of the box
# (the steps of earlier rounds' one-off probes: docs/history/tools/gpu_session_r05.sh)
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
for step in "$@"; do
  echo "== $step"
  case $step in
    stack)   (cat /opt/rocm/.info/version 2>/dev/null; python -c "import torch; print('torch', torch.__version__, 'hip', torch.version.hip, torch.cuda.get_device_name(0))"; rocminfo 2>/dev/null | grep -m3 -i "gfx\|Marketing"; nproc; lscpu | grep -m1 "Model name") > gpurun_out/stack.txt 2>&1; cat gpurun_out/stack.txt ;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log ;;
    tests)   rm -f gpurun_out/reduction_error.jsonl; timeout 3000 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log; python tools/summarize_reduction_error.py > gpurun_out/reduction_error.txt 2>&1; cat gpurun_out/reduction_error.txt ;;
    driver)  tag=${DRIVER_TAG:-1}; t0=$(date +%s%N)
             timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_$tag.json 2> gpurun_out/bench_driver_$tag.err; rc=$?
             t1=$(date +%s%N); echo "driver command rc=$rc wall=$(( (t1 - t0) / 1000000 )) ms lines=$(wc -l < gpurun_out/bench_driver_$tag.json) bytes=$(wc -c < gpurun_out/bench_driver_$tag.json)"
             cp bench_detail.json gpurun_out/bench_detail_$tag.json
             python - <<PYEOF
import json
out = open('gpurun_out/bench_driver_$tag.json').read()
d = json.loads(out[-6000:].splitlines()[-1])          # what the driver does with its record of stdout
r = d['roofline']
print('value', d['value'], 'frac', r['frac'], 'rocprof_frac', r.get('rocprof_frac'), 'traffic', r.get('traffic_over_algorithmic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('kind'))
print('kernel rows', r.get('kernel_rows'), 'worst', r.get('worst_kernel'), r.get('worst_kernel_frac'), '| parity', d.get('parity_bit_exact_vs_reference'), d.get('parity_bit_exact_vs_oracle'))
print('steps/s', d.get('steps_per_sec'))
print('legs', d.get('legs_wall_s'), '| process', d.get('bench_process'))
PYEOF
             ;;
    bench)   timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$? bytes=$(wc -c < gpurun_out/bench.json)"; cp bench_detail.json gpurun_out/bench_detail.json; cat gpurun_out/bench.json ;;
    torchrun) # the way the driver starts N > 1, with one rank: env rendezvous, RCCL group from the launcher's environment
             timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --detail gpurun_out/bench_detail_torchrun.json > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$? bytes=$(wc -c < gpurun_out/bench_torchrun.json)"; tail -1 gpurun_out/bench_torchrun.json | cut -c1-400 ;;
    ranks2)  # the whole multi-rank flow on this one GPU: two ranks on device 0, collectives through gloo (bench.py QD_BENCH_BACKEND)
             QD_BENCH_BACKEND=gloo QD_BENCH_ONE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 20 --warmup 5 --no-kernels --detail gpurun_out/bench_detail_two_ranks_one_gpu.json > gpurun_out/bench_two_ranks_one_gpu.json 2> gpurun_out/bench_two_ranks_one_gpu.err; echo "ranks2 rc=$? bytes=$(wc -c < gpurun_out/bench_two_ranks_one_gpu.json)"; tail -1 gpurun_out/bench_two_ranks_one_gpu.json ;;
    kprof)   # the kernel rows under rocprofv3: the per-kernel average durations next to the HIP-event figures
             rm -rf gpurun_out/kprof_stats
             (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kprof_stats -o kernels -- python $R/tools/bench_kernels.py 26 --no-sweeps > $R/gpurun_out/kernels_under_rocprof.txt 2> $R/gpurun_out/kprof.err); echo "kprof rc=$?"
             find gpurun_out/kprof_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/kernels_rocprof_stats.csv \;
             find gpurun_out/kprof_stats -name '*.csv' -size +4M -delete
             head -12 gpurun_out/kernels_rocprof_stats.csv | cut -c1-200 ;;
    prof)    rm -rf gpurun_out/prof_stats gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
             (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --worker --steps 100 --warmup 10 --no-cpu-baseline --no-distill --no-pmc --no-kernels --detail $R/gpurun_out/prof_bench_detail.json > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "stats rc=$?"
             find gpurun_out/prof_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/bench_kernel_stats.csv \;
             head -4 gpurun_out/bench_kernel_stats.csv | cut -c1-220
             for c in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o bench -- python $R/bench.py --worker --steps 10 --warmup 2 --precondition-s 0.05 --no-cpu-baseline --no-distill --no-pmc --no-kernels --detail /tmp/pmc_detail.json > /dev/null 2> $R/gpurun_out/pmc_$c.err); echo "pmc $c rc=$?"
             done ;;     # (then, back home: python tools/summarize_profiles.py r06)
    kernels) timeout 1200 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/kernels.txt; tail -5 gpurun_out/kernels.txt ;;
    div)     timeout 900 python tools/div_invariant_check.py --pairs 1e9 --cpu 100000 > gpurun_out/div_invariant.txt 2>&1; echo "div rc=$?"; cat gpurun_out/div_invariant.txt ;;
    soak)    QD_SOAK=10 timeout 1500 python -m pytest tests/test_hip_property.py -x -q -m gpu > gpurun_out/property_soak.log 2>&1; tail -2 gpurun_out/property_soak.log ;;
    coverage) timeout 3000 python tools/launch_coverage.py --run > gpurun_out/launch_coverage.log 2>&1; echo "coverage rc=$?"; head -40 gpurun_out/launch_coverage.txt; tail -5 gpurun_out/launch_coverage.log ;;
    dispatch) timeout 1200 python tools/dispatch_map.py --trace > gpurun_out/dispatch_map.log 2>&1; echo "dispatch rc=$?"; head -30 gpurun_out/dispatch_map.txt; tail -3 gpurun_out/dispatch_map.log ;;
    miopen)  timeout 900 python tools/miopen_find_probe.py default 2 > gpurun_out/miopen_find_probe.txt 2>&1; cat gpurun_out/miopen_find_probe.txt ;;
    abk8)    timeout 120 build/ab_k8 > gpurun_out/ab_k8.txt 2>&1; cat gpurun_out/ab_k8.txt ;;
    *)       echo "unknown step $step" ;;
  esac
done
