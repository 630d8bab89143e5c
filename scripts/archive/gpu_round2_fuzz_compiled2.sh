#!/bin/bash
# randomised parity with compiled kernels on the scene classes that reach them since the limits went to 64
# primitives: crowded scenes (17 - 40 primitives), rich ones (stochastic surfaces, gratings, absorbing media,
# sequential mode), crowded scenes with paraboloids -- one hiprtc compile per scene
set -o pipefail
export ODW_COMPILE=structure
mkdir -p gpurun_out
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r2fzc2_progress.log
  timeout -k 10 1100 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a gpurun_out/r2fzc2_progress.log
  [ $rc -le 1 ]
}
run r2fzc2_crowded.log python tests/fuzz_parity.py 40 10000 513 3 &&
run r2fzc2_rich.log python tests/fuzz_parity.py 60 10000 512 1 &&
run r2fzc2_parab_crowded.log python tests/fuzz_parity.py 30 10000 515 5
