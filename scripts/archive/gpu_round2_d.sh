#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== variants c4" | tee gpurun_out/r2d_variants.log
timeout -k 10 900 bash scripts/try_variants.sh --config c4 2>&1 | tee -a gpurun_out/r2d_variants.log
rc=${PIPESTATUS[0]}; [ $rc -le 1 ] || exit $rc
timeout -k 10 900 python scripts/profile_round.py r02a_c4 --config c4 > gpurun_out/r2d_profile_c4.log 2>&1; rc=$?; tail -5 gpurun_out/r2d_profile_c4.log | cut -c1-700; [ $rc -le 1 ]
