#!/bin/bash
# round 5: batch launches against launches of their own on random structures (tests/fuzz_batch.py), generic and compiled kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/r5fb_progress.log
run() {
  local log=$1; shift
  echo "== $*" | tee -a $P
  timeout -k 10 1000 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a $P
  [ $rc -le 1 ]
}
S=${1:-4101}
run r5fb_plain.log python tests/fuzz_batch.py 300 20000 $S 0 &&
run r5fb_parab.log python tests/fuzz_batch.py 150 20000 $((S+1)) 4 &&
run r5fb_rich.log python tests/fuzz_batch.py 400 20000 $((S+2)) 1 &&
ODW_COMPILE=structure run r5fb_plain_compiled.log python tests/fuzz_batch.py 120 20000 $((S+3)) 0 &&
ODW_COMPILE=structure run r5fb_rich_compiled.log python tests/fuzz_batch.py 150 20000 $((S+4)) 1
