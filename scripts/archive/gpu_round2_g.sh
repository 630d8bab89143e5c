#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
step() {
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2g_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2g_progress.log
  tail -n 6 "gpurun_out/$log" | cut -c1-600
  [ $rc -le 1 ]
}
step 300 r2g_bench_c3.log python bench.py --no-cpu-baseline --no-end-to-end &&
step 300 r2g_bench_c4.log python bench.py --config c4 --no-cpu-baseline &&
step 600 r2g_parab_tests.log python -m pytest tests/test_gpu_fuzz.py tests/test_oracle_physics.py tests/test_run_simulation_cpu.py tests/test_gpu_parity_geometry.py -m gpu -q --durations=5 &&
step 300 r2g_fuzz_parab.log python tests/fuzz_parity.py 40 10000 301 5
