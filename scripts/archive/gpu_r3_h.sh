#!/bin/bash
# round 3, call H: mesh kernel -- phase statistics and walk-threshold variants
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
for seg in 64 256 1024; do
  echo "stats seg $seg"
  ODW_GRID_STATS=1 ODW_TRACE_LIB=$PWD/build/libodw_mstats.so timeout -k 10 300 python scripts/bench_mesh.py --segments $seg --steps 1 --warmup 0 2>&1 | tee $O/r03h_stats_$seg.log
done
for v in mstep4 mstep24 mstep40; do
  echo "variant $v"
  ODW_TRACE_LIB=$PWD/build/libodw_$v.so timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03h_err.log | tee $O/r03h_mesh_$v.jsonl
done
