#!/bin/bash
# round 3: slots per hit-list reservation on the headline config (C3, compiled kernel), two passes each
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for pass in 1 2; do
for b in 512 1024 2048; do
  ODW_HIT_BLOCK=$b python bench.py --no-extra --no-cpu-baseline --no-end-to-end > gpurun_out/r03/r03w_c3.json 2>/dev/null
  python - "$b" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r03/r03w_c3.json').read().strip().splitlines()[-1])
print('c3 block %-5s %.4g rays/s  %.3f ms per step  sclk %s' % (sys.argv[1], d['value'], d['ms_per_step'], (d.get('clock') or {}).get('sclk_mhz_mean')))
PY
done
done
