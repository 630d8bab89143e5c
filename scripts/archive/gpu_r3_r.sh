#!/bin/bash
# round 3: every kernel of the C5 sweep on the final library (kernel trace of bench.py --config c5, 1 + 2 sweeps)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
ODW_SWEEP_PIPELINE=${ODW_SWEEP_PIPELINE:-1} rocprofv3 --kernel-trace --stats -d gpurun_out/r03r_c5_trace -o c5 --output-format csv -- python3 bench.py --config c5 --warmup 1 --steps 2 --no-cpu-baseline > gpurun_out/r03/r03r_c5_trace.log 2>&1 || { tail -20 gpurun_out/r03/r03r_c5_trace.log; exit 1; }
f=$(find gpurun_out/r03r_c5_trace -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r03/r03r_c5_sweep_kernel_stats.csv
tail -1 gpurun_out/r03/r03r_c5_trace.log | cut -c1-300
rm -rf gpurun_out/r03r_c5_trace
