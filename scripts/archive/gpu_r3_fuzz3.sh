#!/bin/bash
# round 3: long-form randomised parity on the round's library (device vs oracle, whole trajectories):
# rich scenes (tessellated solids -> mesh kernel, unless stochastic), crowded (grid kernel), plain, paraboloids, sources
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r03/r3fzC_progress.log
  timeout -k 10 900 "$@" > "gpurun_out/r03/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/r03/$log | cut -c1-200)" | tee -a gpurun_out/r03/r3fzC_progress.log
  [ $rc -le 1 ]
}
run r3fzC_rich.log python tests/fuzz_parity.py 500 10000 901 1 &&
run r3fzC_crowded.log python tests/fuzz_parity.py 300 10000 902 3 &&
run r3fzC_plain.log python tests/fuzz_parity.py 300 10000 903 0 &&
run r3fzC_parab_crowded.log python tests/fuzz_parity.py 150 10000 904 5 &&
run r3fzC_sources.log python tests/fuzz_sources.py 100 10000 905 &&
run r3fzC_emitters.log python tests/fuzz_emitters.py 100 50000 906
