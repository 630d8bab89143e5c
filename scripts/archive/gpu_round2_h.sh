#!/bin/bash
# round-2 evidence: full GPU suite, bench lines of every config, profiles of the two dominant kernels
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step() {
  local t=$1 log=$2; shift 2
  echo "== $* (limit ${t}s)" | tee -a gpurun_out/r2h_progress.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/r2h_progress.log
  tail -n 4 "gpurun_out/$log" | cut -c1-300
  [ $rc -le 1 ]
}
step 900 r2h_all_gpu_tests.log python -m pytest tests -m gpu -q --durations=10 &&
step 300 r2h_smoke.log python __graft_entry__.py smoke &&
step 600 r2h_profile_c3.log python scripts/profile_round.py r02c_c3 --config c3 &&
step 600 r2h_profile_c4.log python scripts/profile_round.py r02c_c4 --config c4 &&
step 600 r2h_bench_c5.log python bench.py --config c5 &&
step 600 r2h_configs.log python scripts/bench_configs.py
