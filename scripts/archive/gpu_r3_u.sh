#!/bin/bash
# round 3: the grid kernel's walk threshold again, after the round's changes (C4, 3 steps each)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for v in "" step16 step32 step40; do
  if [ -n "$v" ]; then export ODW_TRACE_LIB=$PWD/build/libodw_$v.so; fi
  python bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03/r03u_c4.json 2>/dev/null
  python - "$v" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r03/r03u_c4.json').read().strip().splitlines()[-1])
print('c4 %-8s %.4g rays/s  %.3f ms per step' % (sys.argv[1] or 'step24', d['value'], d['ms_per_step']))
PY
done
