#!/bin/bash
# round 3: blocks per CU of the flat kernels' grid (8 against the 4 that are resident): C3, one C5 launch size, the sweep
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for pass in 1 2; do
for m in 8 4; do
  ODW_GRID_MULT=$m python bench.py --no-extra --no-cpu-baseline --no-end-to-end > gpurun_out/r03/r03x_c3.json 2>/dev/null
  ODW_GRID_MULT=$m python bench.py --config c5 --warmup 1 --steps 3 --no-cpu-baseline > gpurun_out/r03/r03x_c5.json 2>/dev/null
  python - "$m" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r03/r03x_c3.json').read().strip().splitlines()[-1])
e=json.loads(open('gpurun_out/r03/r03x_c5.json').read().strip().splitlines()[-1])
print('mult %s  c3 %.3f ms per step   c5 %.1f ms per sweep, kernel %.4f ms per radius' % (sys.argv[1], d['ms_per_step'], e['ms_per_step'], e['roofline']['avg_kernel_ms']))
PY
done
done
ODW_SL_SIZES=1e6,3e6,1e7,3e7 ODW_SL_REPS=20 ODW_GRID_MULT=8 python scripts/short_launch.py
ODW_SL_SIZES=1e6,3e6,1e7,3e7 ODW_SL_REPS=20 ODW_GRID_MULT=4 python scripts/short_launch.py
