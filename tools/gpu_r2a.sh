#!/bin/bash
# round 2, session A: new tests + bench with the new legs
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 2400 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 1200 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
