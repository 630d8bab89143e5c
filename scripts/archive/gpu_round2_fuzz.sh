#!/bin/bash
# long-form randomised parity on the final library (device vs oracle, whole trajectories)
set -o pipefail
mkdir -p gpurun_out
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r2fzF_progress.log
  timeout -k 10 900 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a gpurun_out/r2fzF_progress.log
  [ $rc -le 1 ]
}
run r2fzF_plain.log python tests/fuzz_parity.py 200 10000 411 0 &&
run r2fzF_rich.log python tests/fuzz_parity.py 150 10000 412 1 &&
run r2fzF_crowded.log python tests/fuzz_parity.py 200 10000 413 3 &&
run r2fzF_parab.log python tests/fuzz_parity.py 150 10000 414 4 &&
run r2fzF_parab_crowded.log python tests/fuzz_parity.py 150 10000 415 5 &&
run r2fzF_sources.log python tests/fuzz_sources.py 100 10000 416 &&
run r2fzF_emitters.log python tests/fuzz_emitters.py 150 50000 417
