#!/bin/bash
# round 3, call C: whole GPU suite (explicit rays component-major, pinned column pool), run-loop throughput
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03c_tests.log 2>&1 || { tail -40 gpurun_out/r03c_tests.log; exit 1; }
tail -3 gpurun_out/r03c_tests.log
python scripts/bench_run_simulation.py 1e8 > gpurun_out/r03c_run_loop.log 2>&1 || { tail -30 gpurun_out/r03c_run_loop.log; exit 1; }
grep '^{' gpurun_out/r03c_run_loop.log
python scripts/bench_run_simulation.py 1e8 16777216 > gpurun_out/r03c_run_loop16.log 2>&1 || { tail -30 gpurun_out/r03c_run_loop16.log; exit 1; }
grep '^{' gpurun_out/r03c_run_loop16.log
