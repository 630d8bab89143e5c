#!/bin/bash
# round 5: after cones and the deeper heuristic -- leaf size, walk threshold
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_mesh_scan.log
: > $O
export ODW_BVH_SAH_DEEP=1
for L in 8 4 6 12 15; do
  echo "== leaf $L" | tee -a $O
  ODW_BVH_LEAF=$L timeout -k 10 300 python scripts/bench_mesh.py --segments 256 1024 2>&1 | cut -c1-330 | tee -a $O || exit 1
done
for V in meshstep8 meshstep24 meshstep32; do
  echo "== variant $V" | tee -a $O
  ODW_TRACE_LIB=build/libodw_$V.so timeout -k 10 300 python scripts/bench_mesh.py --segments 256 1024 2>&1 | cut -c1-330 | tee -a $O || exit 1
done
