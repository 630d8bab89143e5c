#!/bin/bash
# first GPU pass of round 2: new tests, bench lines of every config, counter list
# (a step that times out or dies stops the chain; an ordinary test failure does not)
set -o pipefail
mkdir -p gpurun_out
step() {  # step <seconds> <log> <cmd...>
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2a_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2a_progress.log
  tail -n 5 "gpurun_out/$log"
  [ $rc -le 1 ]
}
step 900 r2a_new_tests.log python -m pytest tests/test_gpu_multi.py tests/test_oracle_physics.py tests/test_fan_notebook.py tests/test_run_simulation_cpu.py -m gpu -q --durations=10 &&
step 600 r2a_bench_c3.log python bench.py &&
step 600 r2a_bench_c4.log python bench.py --config c4 &&
step 900 r2a_bench_c5.log python bench.py --config c5 &&
step 900 r2a_scale.log python -m pytest tests/test_gpu_scale.py -m gpu -q --durations=10 &&
step 120 r2a_counters.txt rocprofv3 -L
