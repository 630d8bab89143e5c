#!/bin/bash
# round 3: randomised parity, the classes the last campaigns left out: rich + crowded, plain with paraboloids, and plain /
# rich scenes through the scene-compiled kernels (ODW_COMPILE=structure: every tracer compiles its scene)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a gpurun_out/r03/r3fzF_progress.log
  timeout -k 10 900 "$@" > "gpurun_out/r03/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/r03/$log | cut -c1-200)" | tee -a gpurun_out/r03/r3fzF_progress.log
  [ $rc -le 1 ]
}
run r3fzF_rich_crowded.log python tests/fuzz_parity.py 300 10000 1201 2 &&
run r3fzF_parab.log python tests/fuzz_parity.py 300 10000 1202 4 &&
ODW_COMPILE=structure run r3fzF_plain_compiled.log python tests/fuzz_parity.py 150 10000 1203 0 &&
ODW_COMPILE=structure run r3fzF_rich_compiled.log python tests/fuzz_parity.py 150 10000 1204 1
