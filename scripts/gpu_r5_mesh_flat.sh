#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_mesh_flat.log
: > $O
timeout -k 10 900 python -m pytest tests/test_mesh.py tests/test_gpu_fuzz.py -m gpu -x -q -k "mesh or cones or lopsided or single_leaf or presorted or segment" 2>&1 | tail -3 | tee -a $O || exit 1
for F in 1 0; do
  echo "== ODW_MESH_FLAT=$F" | tee -a $O
  export ODW_MESH_FLAT=$F
  timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>&1 | cut -c1-330 | tee -a $O || exit 1
  timeout -k 10 300 python scripts/bench_mesh.py --segments 1024 --sigma 0.12 2>&1 | cut -c1-330 | tee -a $O || exit 1
  timeout -k 10 300 python scripts/bench_facet_scenes.py 2>&1 | tail -3 | cut -c1-300 | tee -a $O
done
ODW_MESH_FLAT=1 ODW_GRID_STATS=1 ODW_TRACE_LIB=build/libodw_meshstats.so timeout -k 10 300 python scripts/bench_mesh.py --segments 1024 --steps 1 --warmup 0 --plain 2>&1 | grep -v "^{" | cut -c1-200 | tee -a $O || exit 1
