#!/bin/bash
# round 3, call K: mesh kernel variants (build/libodw_*.so), leaf 4
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
for lib in build/libodw_*.so; do
  v=$(basename $lib .so)
  echo "variant $v"
  ODW_TRACE_LIB=$PWD/$lib timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03k_err.log | tee $O/r03k_$v.jsonl
done
