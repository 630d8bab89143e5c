#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
step() {
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2e_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2e_progress.log
  tail -n 8 "gpurun_out/$log" | cut -c1-500
  [ $rc -le 1 ]
}
step 180 r2e_parity.log python -m pytest tests/test_gpu_parity.py tests/test_trace_golden.py tests/test_gpu_scale.py -m gpu -q -x --durations=3 &&
step 300 r2e_fuzz_crowded.log python tests/fuzz_parity.py 30 10000 202 3 &&
ODW_GRID_STATS=1 ODW_TRACE_LIB=$PWD/build/libodw_stats.so step 200 r2e_stats.log python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline &&
rm -f build/libodw_stats.so && (timeout -k 10 600 bash scripts/try_variants.sh --config c4 2>&1 | tee gpurun_out/r2e_variants.log)
