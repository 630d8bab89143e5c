#!/bin/bash
# round 3: long-form randomised parity on the round's library (device vs oracle, whole trajectories):
# rich scenes (tessellated solids -> mesh kernel, unless stochastic), crowded (grid kernel), plain, paraboloids, sources
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r03/r3fzB_progress.log
  timeout -k 10 900 "$@" > "gpurun_out/r03/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/r03/$log | cut -c1-200)" | tee -a gpurun_out/r03/r3fzB_progress.log
  [ $rc -le 1 ]
}
run r3fzB_rich.log python tests/fuzz_parity.py 400 10000 711 1 &&
run r3fzB_crowded.log python tests/fuzz_parity.py 200 10000 712 3 &&
run r3fzB_plain.log python tests/fuzz_parity.py 150 10000 713 0 &&
run r3fzB_parab_crowded.log python tests/fuzz_parity.py 100 10000 714 5 &&
run r3fzB_sources.log python tests/fuzz_sources.py 100 10000 715 &&
run r3fzB_emitters.log python tests/fuzz_emitters.py 100 50000 716
