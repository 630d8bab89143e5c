#!/bin/bash
# round 3: per-block counter atomics: short launches (generic + compiled kernel), the GPU suite, the default bench line
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
T=${1:-r03p}
ODW_SL_COMPILE=off python scripts/short_launch.py > $O/${T}_short_off.log 2>&1 || { tail -20 $O/${T}_short_off.log; exit 1; }
cat $O/${T}_short_off.log
python scripts/short_launch.py > $O/${T}_short_spec.log 2>&1 || { tail -20 $O/${T}_short_spec.log; exit 1; }
cat $O/${T}_short_spec.log
bash scripts/gpu_r3_final.sh $T
