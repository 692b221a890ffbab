#!/bin/bash
# One MI355X lease, driven by a list of step names:   gpurun -- 'bash tools/gpu_session.sh smoke tests bench'
# Every step writes under gpurun_out/ (merged back by gpurun); tools/summarize_*.py turn the logs into profiles/r04_*.
#   smoke      __graft_entry__.smoke()
#   tests      the whole GPU suite (records the achieved reduction errors in gpurun_out/reduction_error.jsonl)
#   bench      bench.py with the default flags, then with the driver's (--steps 20 --warmup 5)
#   prof       rocprofv3 --kernel-trace --stats of the bench command (its measuring process, `bench.py --worker`, run directly) + the two PMC passes (FETCH_SIZE, WRITE_SIZE)
#   kernels    tools/bench_kernels.py (every kernel of the path, steady state)
#   pmck       per-kernel PMC traffic of tools/pmc_probe.py
#   div        tools/div_invariant_check.py --pairs 1e9 (the long run of the division proof)
#   ragged     tools/ragged_probe.py
#   api        tools/profile_api_overhead.py
#   soak       property tests with QD_SOAK=10
#   sqk6       SQ counters of the point-gradient kernels (k = 16 / 64 / 128 / 256; tools/sq_probe_k6.py)
#   sq         SQ counters (VALU / LDS instructions, LDS bank conflicts) of the nearest-point calls, before / after
#   side       tools/side_output_probe.py (calls with index / level side outputs at every bucket-size family; SIDE_ARGS)
#   spread     tools/distill_spread_probe.py: repetition-to-repetition spread of the configs[1] step, with a kernel trace
#   stack      ROCm / driver / torch versions of the box
#   torchrun   bench.py under torch.distributed.run with one rank;  ranks2: two ranks on this one GPU through gloo
#   kprof      tools/bench_kernels.py --no-sweeps under rocprofv3 --kernel-trace --stats
#   coverage   tools/launch_coverage.py --run: the GPU suite under rocprofv3 --kernel-trace --stats, shipped kernels never launched
#   dispatch   tools/dispatch_map.py --trace: call geometry -> kernel map
#   driver     the driver's exact command (python3 bench.py --gpus 1 --steps 20 --warmup 5), output numbered by DRIVER_TAG (soak: one per lease)
#   abk6m      tools/ab_k6m.py: the multi-tensor point-gradient sweep, this build against build/libqd_hip_prev.so
#   capture    tests/test_hip_capture_watchdog.py, then the round-4 configuration on purpose (global-mode capture, collectives in flight)
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
    bench)   timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/bench.json
             timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_flags.json 2> gpurun_out/bench_driver_flags.err; python -c "import json; d=json.loads(open('gpurun_out/bench_driver_flags.json').read().strip().splitlines()[-1]); print('driver flags:', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], len(d['roofline'].get('kernels') or []), 'kernel rows')" ;;
    torchrun) # the way the driver starts N > 1, with one rank: env rendezvous, RCCL group from the launcher's environment
             timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$?"; tail -1 gpurun_out/bench_torchrun.json | cut -c1-300 ;;
    ranks2)  # the whole multi-rank flow on this one GPU: two ranks on device 0, collectives through gloo (bench.py QD_BENCH_BACKEND)
             QD_BENCH_BACKEND=gloo QD_BENCH_ONE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-kernels > gpurun_out/bench_two_ranks_one_gpu.json 2> gpurun_out/bench_two_ranks_one_gpu.err; echo "ranks2 rc=$?"; tail -1 gpurun_out/bench_two_ranks_one_gpu.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['n_gpus'], d['collective_backend']); [print(k, r[k]) for k in r if k.startswith('dp_') or k.startswith('steps_')]" ;;
    kprof)   # the kernel rows under rocprofv3: the per-kernel average durations next to the HIP-event figures
             rm -rf gpurun_out/kprof_stats
             (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kprof_stats -o kernels -- python $R/tools/bench_kernels.py 26 --no-sweeps > $R/gpurun_out/kernels_under_rocprof.txt 2> $R/gpurun_out/kprof.err); echo "kprof rc=$?"
             find gpurun_out/kprof_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/kernels_rocprof_stats.csv \;
             find gpurun_out/kprof_stats -name '*.csv' -size +4M -delete
             head -12 gpurun_out/kernels_rocprof_stats.csv | cut -c1-200 ;;
    prof)    rm -rf gpurun_out/prof_stats gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
             (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --worker --steps 100 --warmup 10 --no-cpu-baseline --no-distill --no-pmc --no-kernels > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err); echo "stats rc=$?"
             for c in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o bench -- python $R/bench.py --worker --steps 10 --warmup 2 --precondition-s 0.05 --no-cpu-baseline --no-distill --no-pmc --no-kernels > /dev/null 2> $R/gpurun_out/pmc_$c.err); echo "pmc $c rc=$?"
             done ;;
    pmck)    rm -rf gpurun_out/pmcK_FETCH_SIZE gpurun_out/pmcK_WRITE_SIZE
             for c in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcK_$c -o probe -- python $R/tools/pmc_probe.py > /dev/null 2> $R/gpurun_out/pmcK_$c.err); echo "pmcK $c rc=$?"
             done ;;
    kernels) timeout 1200 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/kernels.txt; tail -5 gpurun_out/kernels.txt ;;
    div)     timeout 900 python tools/div_invariant_check.py --pairs 1e9 --cpu 100000 > gpurun_out/div_invariant.txt 2>&1; echo "div rc=$?"; cat gpurun_out/div_invariant.txt ;;
    ragged)  timeout 600 python tools/ragged_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ragged.txt; cat gpurun_out/ragged.txt ;;
    api)     timeout 600 python tools/profile_api_overhead.py 2>&1 | grep -v amdgpu.ids > gpurun_out/api_overhead.txt; head -8 gpurun_out/api_overhead.txt ;;
    soak)    QD_SOAK=10 timeout 1500 python -m pytest tests/test_hip_property.py -x -q -m gpu > gpurun_out/property_soak.log 2>&1; tail -2 gpurun_out/property_soak.log ;;
    sq)      # SQ counters of the nearest-point calls, current library and (when present) build/libqd_hip_prev.so; dispatch order
             # inside tools/sq_probe_r3.py: K5 bucket 256 k = 256 / 16 / 4, bucket 100 k = 4, bucket 33 k = 4, bucket 100 k = 256
             for tag in cur prev; do
               [ $tag = prev ] && { [ -f build/libqd_hip_prev.so ] || continue; export QD_LIB=$R/build/libqd_hip_prev.so; }
               rm -rf gpurun_out/sq_$tag
               (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/sq_$tag -o sq -- python $R/tools/sq_probe_r3.py > /dev/null 2> $R/gpurun_out/sq_$tag.err); echo "sq $tag rc=$?"
               unset QD_LIB
             done
             python tools/sq_summarize_r3.py > gpurun_out/sq_counters.txt 2>&1; cat gpurun_out/sq_counters.txt ;;
    sqk6)    # SQ counters of the point-gradient kernels (tools/sq_probe_k6.py)
             rm -rf gpurun_out/sq_k6
             (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/sq_k6 -o sq -- python $R/tools/sq_probe_k6.py > /dev/null 2> $R/gpurun_out/sq_k6.err); echo "sq k6 rc=$?"
             python tools/sq_probe_k6.py --summarize > gpurun_out/sq_counters_k6.txt 2>&1; cat gpurun_out/sq_counters_k6.txt
             find gpurun_out/sq_k6 -name '*.csv' -size +8M -delete ;;
    side)    timeout 900 python tools/side_output_probe.py $SIDE_ARGS 2>&1 | grep -v amdgpu.ids > gpurun_out/side_output.txt; cat gpurun_out/side_output.txt ;;
    spread)  timeout 600 python tools/distill_spread_probe.py --sleep 0.5 2>&1 | grep -v amdgpu.ids > gpurun_out/distill_spread.txt; head -20 gpurun_out/distill_spread.txt
             rm -rf gpurun_out/spread_trace
             (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/spread_trace -o spread -- python $R/tools/distill_spread_probe.py --marks --reps 8 > $R/gpurun_out/distill_spread_traced.txt 2> $R/gpurun_out/spread.err); echo "trace rc=$?"
             python tools/distill_spread_probe.py --analyse gpurun_out/spread_trace > gpurun_out/distill_spread_analysis.txt 2>&1; cat gpurun_out/distill_spread_analysis.txt
             find gpurun_out/spread_trace -name '*.csv' -size +8M -delete ;;
    coverage) timeout 3000 python tools/launch_coverage.py --run > gpurun_out/launch_coverage.log 2>&1; echo "coverage rc=$?"; head -40 gpurun_out/launch_coverage.txt; tail -5 gpurun_out/launch_coverage.log ;;
    dispatch) timeout 1200 python tools/dispatch_map.py --trace > gpurun_out/dispatch_map.log 2>&1; echo "dispatch rc=$?"; head -30 gpurun_out/dispatch_map.txt; tail -3 gpurun_out/dispatch_map.log ;;
    driver)  tag=${DRIVER_TAG:-1}; t0=$(date +%s%N)
             timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_$tag.json 2> gpurun_out/bench_driver_$tag.err; rc=$?
             t1=$(date +%s%N); echo "driver command rc=$rc wall=$(( (t1 - t0) / 1000000 )) ms lines=$(wc -l < gpurun_out/bench_driver_$tag.json)"
             python - <<PYEOF
import json
d = json.loads(open('gpurun_out/bench_driver_$tag.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'frac', r['frac'], 'rocprof_frac', r.get('rocprof_frac'), 'traffic', r.get('traffic_over_algorithmic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('kind'))
print('kernel rows', len(r.get('kernels') or []), '| parity', d.get('parity_bit_exact_vs_reference'), d.get('parity_bit_exact_vs_oracle'))
print('legs', r.get('legs_wall_s'), '| wall', r.get('wall_s'))
print('process', d.get('bench_process'))
for k in r:
    if k.startswith('steps_') or k.startswith('dp_'):
        print(k, r[k])
PYEOF
             ;;
    capture) timeout 1500 python -m pytest tests/test_hip_capture_watchdog.py -q -m gpu > gpurun_out/capture_tests.log 2>&1; echo "capture tests rc=$?"; tail -3 gpurun_out/capture_tests.log
             # what round 4 ran, provoked: global-mode capture with collectives in flight -- expected to die (SIGABRT = rc 134)
             timeout 600 python tests/capture_worker.py --mode global --settle 0 --inflight 8 --iters 20 > gpurun_out/capture_global_mode.log 2>&1; echo "global-mode capture, collectives in flight: rc=$?"
             grep -m3 "capturing\|CAPTURE_OK\|terminate" gpurun_out/capture_global_mode.log | cut -c1-300
             timeout 600 python tests/capture_worker.py --mode global --settle 0.35 --iters 20 > gpurun_out/capture_global_mode_settled.log 2>&1; echo "global-mode capture after quiescing: rc=$?"
             grep -m3 "capturing\|CAPTURE_OK\|terminate" gpurun_out/capture_global_mode_settled.log | cut -c1-300 ;;
    abk6m)   timeout 900 python tools/ab_k6m.py ${ABK6M_ARGS:-4 16 64} 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_k6m.txt; cat gpurun_out/ab_k6m.txt
             # per-kernel durations (sweep and fold separately), per shape list
             for m in wrn one64Mi; do
               rm -rf gpurun_out/abk6m_stats
               (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk6m_stats -o ab -- python $R/tools/ab_k6m.py 4 --models=$m > /dev/null 2> $R/gpurun_out/abk6m_prof.err); echo "abk6m rocprof $m rc=$?"
               find gpurun_out/abk6m_stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/ab_k6m_kernel_stats_$m.csv \;
               grep "point_grad" gpurun_out/ab_k6m_kernel_stats_$m.csv | cut -c1-200
             done
             rm -rf gpurun_out/abk6m_stats ;;
    *)       echo "unknown step $step" ;;
  esac
done
