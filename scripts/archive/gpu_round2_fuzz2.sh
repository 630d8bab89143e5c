#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() {
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r2fz2_progress.log
  timeout -k 10 900 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a gpurun_out/r2fz2_progress.log
  [ $rc -le 1 ]
}
run r2fz2_rich.log python tests/fuzz_parity.py 150 10000 212 1 &&
run r2fz2_rich2.log python tests/fuzz_parity.py 150 10000 218 2 &&
run r2fz2_sources.log python tests/fuzz_sources.py 100 10000 216
