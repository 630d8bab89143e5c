#!/bin/bash
# round 4: randomised device-vs-oracle parity on the round's final library (batch-capable flat kernels, min / max
# forms, hit-list blocks without wasted slots, mesh kernel with sorted hand-out order for generated rays)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/r4fz_progress.log
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a $P
  timeout -k 10 1000 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a $P
  [ $rc -le 1 ]
}
S=${1:-1401}
run r4fz_plain.log python tests/fuzz_parity.py 250 10000 $S 0 &&
run r4fz_rich.log python tests/fuzz_parity.py 300 10000 $((S+1)) 1 &&
run r4fz_crowded.log python tests/fuzz_parity.py 200 10000 $((S+2)) 3 &&
run r4fz_parab.log python tests/fuzz_parity.py 150 10000 $((S+3)) 5 &&
ODW_COMPILE=structure run r4fz_plain_compiled.log python tests/fuzz_parity.py 150 10000 $((S+4)) 0 &&
run r4fz_sources.log python tests/fuzz_sources.py 100 10000 $((S+5)) &&
ODW_MESH_PRESORT_MIN=1 run r4fz_sources_sorted.log python tests/fuzz_sources.py 60 10000 $((S+6)) &&
run r4fz_emitters.log python tests/fuzz_emitters.py 100 50000 $((S+7))
