#!/bin/bash
# round 3: the mesh kernel at 2 / 3 / 4 waves per SIMD on the final kernel (ball lens, 6.5e4 and 1.05e6 facets)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
O=gpurun_out/r03/r03v_mesh_waves.log
: > $O
for v in "" mw2 mw4; do
  if [ -n "$v" ]; then export ODW_TRACE_LIB=$PWD/build/libodw_$v.so; fi
  echo "== ${v:-mw3}" >> $O
  timeout -k 10 300 python scripts/bench_mesh.py --segments 256 1024 --steps 3 --warmup 1 >> $O 2>&1
done
cat $O
