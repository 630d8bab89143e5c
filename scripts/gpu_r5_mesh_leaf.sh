#!/bin/bash
# round 5: the mesh kernel against the size of the leaves of its tree (ODW_BVH_LEAF): candidates per visit and time;
# interaction inlined / out of line / four waves per SIMD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_mesh_leaf.log
: > $O
timeout -k 10 600 python -m pytest tests/test_mesh.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O || exit 1
for V in meshinl meshw4; do
  echo "== variant $V" | tee -a $O
  ODW_TRACE_LIB=build/libodw_$V.so timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>&1 | cut -c1-330 | tee -a $O || exit 1
done
for L in ${LEAVES:-8 1 2 3 4 6}; do
  echo "== leaf $L" | tee -a $O
  ODW_BVH_LEAF=$L timeout -k 10 300 python scripts/bench_mesh.py --segments ${SEGS:-64 256 1024} 2>&1 | cut -c1-330 | tee -a $O || exit 1
  ODW_BVH_LEAF=$L ODW_GRID_STATS=1 ODW_TRACE_LIB=build/libodw_meshstats.so timeout -k 10 300 python scripts/bench_mesh.py --segments 1024 --steps 1 --warmup 0 --plain 2>&1 | grep -v "^{" | cut -c1-200 | tee -a $O || exit 1
done
