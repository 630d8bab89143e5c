#!/bin/bash
# second GPU pass: device-side Hits, streaming fetch, configs
set -o pipefail
mkdir -p gpurun_out
step() {
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2b_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2b_progress.log
  tail -n 6 "gpurun_out/$log"
  [ $rc -le 1 ]
}
step 600 r2b_device_hits.log python -m pytest tests/test_gpu_device_hits.py tests/test_gpu_multi.py -m gpu -q --durations=10 -x &&
step 600 r2b_bench_c3.log python bench.py --no-cpu-baseline &&
step 600 r2b_bench_c5.log python bench.py --config c5 --no-cpu-baseline &&
step 900 r2b_scale.log python -m pytest tests/test_gpu_scale.py -m gpu -q --durations=10
