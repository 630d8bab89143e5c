#!/bin/bash
# round 5: randomised device-vs-oracle parity on the library with the points table of batch launches, staged uploads,
# the device-resident post-hoc chain and the tabled polar binning
set -o pipefail
cd "$GRAFT_REPO_ROOT"
P=gpurun_out/r5fz_progress.log
run() {  # run <log> <args...>
  local log=$1; shift
  echo "== $*" | tee -a $P
  timeout -k 10 1000 "$@" > "gpurun_out/$log" 2>&1; local rc=$?
  echo "   rc=$rc $(tail -n 1 gpurun_out/$log | cut -c1-200)" | tee -a $P
  [ $rc -le 1 ]
}
S=${1:-1401}
run r5fz_plain.log python tests/fuzz_parity.py 250 10000 $S 0 &&
run r5fz_rich.log python tests/fuzz_parity.py 300 10000 $((S+1)) 1 &&
run r5fz_crowded.log python tests/fuzz_parity.py 200 10000 $((S+2)) 3 &&
run r5fz_parab.log python tests/fuzz_parity.py 150 10000 $((S+3)) 5 &&
ODW_COMPILE=structure run r5fz_plain_compiled.log python tests/fuzz_parity.py 150 10000 $((S+4)) 0 &&
run r5fz_sources.log python tests/fuzz_sources.py 100 10000 $((S+5)) &&
ODW_MESH_PRESORT_MIN=1 run r5fz_sources_sorted.log python tests/fuzz_sources.py 60 10000 $((S+6)) &&
run r5fz_emitters.log python tests/fuzz_emitters.py 100 50000 $((S+7))
