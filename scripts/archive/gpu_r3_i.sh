#!/bin/bash
# round 3, call I: mesh kernel on the eight-wide tree -- parity, rates, occupancy variants, phase statistics
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_mesh.py tests/test_brep.py tests/test_gpu_fuzz.py tests/test_surface_source.py -m gpu -x -q > $O/r03i_tests.log 2>&1 || { tail -60 $O/r03i_tests.log; exit 1; }
tail -2 $O/r03i_tests.log
echo "default"
timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03i_err.log | tee $O/r03i_mesh.jsonl
timeout -k 10 300 python scripts/bench_mesh.py --segments 256 1024 --sigma 0.12 2>$O/r03i_err.log | tee $O/r03i_meshwide.jsonl
for v in mw3; do
  echo "variant $v"
  ODW_TRACE_LIB=$PWD/build/libodw_$v.so timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03i_err.log | tee $O/r03i_mesh_$v.jsonl
done
for seg in 256 1024; do
  echo "stats seg $seg"
  ODW_GRID_STATS=1 ODW_TRACE_LIB=$PWD/build/libodw_mstats.so timeout -k 10 300 python scripts/bench_mesh.py --segments $seg --steps 1 --warmup 0 2>&1 | tee $O/r03i_stats_$seg.log
done
