#!/bin/bash
# round 3: long-form randomised parity on the round's library (device vs oracle, whole trajectories):
# rich scenes (tessellated solids -> mesh kernel, unless stochastic), crowded (grid kernel), plain, paraboloids, sources
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r03/r3fzE_progress.log
  timeout -k 10 900 "$@" > "gpurun_out/r03/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/r03/$log | cut -c1-200)" | tee -a gpurun_out/r03/r3fzE_progress.log
  [ $rc -le 1 ]
}
run r3fzE_rich.log python tests/fuzz_parity.py 700 10000 1101 1 &&
run r3fzE_crowded.log python tests/fuzz_parity.py 400 10000 1102 3 &&
run r3fzE_plain.log python tests/fuzz_parity.py 400 10000 1103 0 &&
run r3fzE_parab_crowded.log python tests/fuzz_parity.py 200 10000 1104 5 &&
run r3fzE_sources.log python tests/fuzz_sources.py 100 10000 1105 &&
run r3fzE_emitters.log python tests/fuzz_emitters.py 100 50000 1106
