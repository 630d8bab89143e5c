#!/bin/bash
# round 3, call G: mesh kernel -- parity (mesh tests, brep, fuzz), then rates against the BVH kernels
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_mesh.py tests/test_brep.py tests/test_gpu_fuzz.py tests/test_surface_source.py -m gpu -x -q > $O/r03g_tests.log 2>&1 || { tail -40 $O/r03g_tests.log; exit 1; }
tail -2 $O/r03g_tests.log
for mode in 1 0; do
  echo "ODW_MESH_KERNEL=$mode"
  ODW_MESH_KERNEL=$mode timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03g_err.log | tee $O/r03g_mesh_k$mode.jsonl
  ODW_MESH_KERNEL=$mode timeout -k 10 300 python scripts/bench_mesh.py --segments 256 1024 --sigma 0.12 2>$O/r03g_err.log | tee $O/r03g_meshwide_k$mode.jsonl
done
