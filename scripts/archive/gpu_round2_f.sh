#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step() {
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2f_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2f_progress.log
  tail -n 6 "gpurun_out/$log" | cut -c1-400
  [ $rc -le 1 ]
}
(timeout -k 10 400 bash scripts/try_variants.sh 2>&1 | tee gpurun_out/r2f_variants_c3.log) &&
rm -f build/*.so &&
step 900 r2f_all_gpu_tests.log python -m pytest tests -m gpu -q --durations=15 &&
step 600 r2f_profile_c4.log python scripts/profile_round.py r02b_c4 --config c4 &&
step 600 r2f_profile_c3.log python scripts/profile_round.py r02b_c3 --config c3
