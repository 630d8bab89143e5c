#!/bin/bash
# grid kernel: parity first (short timeouts: a new persistent kernel), then throughput
set -o pipefail
mkdir -p gpurun_out
step() {
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2c_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2c_progress.log
  tail -n 8 "gpurun_out/$log" | cut -c1-600
  [ $rc -le 1 ]
}
step 180 r2c_parity_huge.log python -m pytest tests/test_gpu_parity.py tests/test_trace_golden.py -m gpu -q -x --durations=5 &&
step 300 r2c_fuzz_crowded.log python tests/fuzz_parity.py 40 10000 201 3 &&
step 300 r2c_bench_c4.log python bench.py --config c4 --no-cpu-baseline &&
ODW_NO_GRID=1 step 300 r2c_bench_c4_bvh.log python bench.py --config c4 --no-cpu-baseline &&
step 300 r2c_gpu_tests_misc.log python -m pytest tests/test_gpu_device_hits.py tests/test_gpu_scale.py tests/test_gpu_errors.py -m gpu -q --durations=5
