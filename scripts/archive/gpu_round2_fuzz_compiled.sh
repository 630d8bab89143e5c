#!/bin/bash
# randomised parity with every tracer compiling its scenes (ODW_COMPILE=structure): device vs oracle on
# whole trajectories, one hiprtc compile per random scene structure
set -o pipefail
export ODW_COMPILE=structure
mkdir -p gpurun_out
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r2fzc_progress.log
  timeout -k 10 1000 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a gpurun_out/r2fzc_progress.log
  [ $rc -le 1 ]
}
run r2fzc_plain.log python tests/fuzz_parity.py 200 10000 311 0 &&
run r2fzc_sources.log python tests/fuzz_sources.py 100 10000 316
