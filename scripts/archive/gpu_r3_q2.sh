#!/bin/bash
# round 3: shrinking hand-out units of the flat kernels: smallest unit, largest unit (compiled kernel, GettingStarted)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
O=gpurun_out/r03/r03q_chunks.log
export ODW_SL_SIZES=1e6,3e6,1e7,3e7,1e8
echo "default (min 128, max 1024)" > $O; python scripts/short_launch.py >> $O 2>&1
echo "fixed units (as before)" >> $O; ODW_CHUNK_MIN=4096 python scripts/short_launch.py >> $O 2>&1
for m in 64 256; do echo "min $m" >> $O; ODW_CHUNK_MIN=$m python scripts/short_launch.py >> $O 2>&1; done
for m in 2048; do echo "max $m" >> $O; ODW_CHUNK_MAX=$m python scripts/short_launch.py >> $O 2>&1; done
echo "generic kernel, default" >> $O; ODW_SL_COMPILE=off python scripts/short_launch.py >> $O 2>&1
echo "generic kernel, fixed" >> $O; ODW_CHUNK_MIN=4096 ODW_SL_COMPILE=off python scripts/short_launch.py >> $O 2>&1
cat $O
python bench.py --no-extra --no-cpu-baseline --no-end-to-end > gpurun_out/r03/r03q_c3.json 2>/dev/null
ODW_CHUNK_MIN=4096 python bench.py --no-extra --no-cpu-baseline --no-end-to-end > gpurun_out/r03/r03q_c3_fixed.json 2>/dev/null
ODW_CHUNK_MAX=2048 python bench.py --no-extra --no-cpu-baseline --no-end-to-end > gpurun_out/r03/r03q_c3_max2048.json 2>/dev/null
python - <<'PY'
import json
for t in ('', '_fixed', '_max2048'):
  d=json.loads(open('gpurun_out/r03/r03q_c3%s.json' % t).read().strip().splitlines()[-1])
  print('c3', t, '%.4g' % d['value'], d['ms_per_step'])
PY
