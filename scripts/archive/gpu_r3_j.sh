#!/bin/bash
# round 3, call J: mesh kernel phase times, leaf sizes
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
for leaf in 8 4 2; do
  echo "leaf $leaf"
  ODW_BVH_LEAF=$leaf timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03j_err.log | tee $O/r03j_mesh_leaf$leaf.jsonl
done
for seg in 256; do
  for leaf in 8 4; do
  echo "stats seg $seg leaf $leaf"
  ODW_BVH_LEAF=$leaf ODW_GRID_STATS=1 ODW_TRACE_LIB=$PWD/build/libodw_mstats.so timeout -k 10 300 python scripts/bench_mesh.py --segments $seg --steps 1 --warmup 0 2>&1 | tee $O/r03j_stats_${seg}_$leaf.log
  done
done
