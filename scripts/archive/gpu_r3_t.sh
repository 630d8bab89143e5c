#!/bin/bash
# round 3: what recording costs in C3 (compiled kernel): rows + histogram / histogram only / rows only
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for v in "" "--no-hits" "--no-histogram"; do
  python bench.py --no-extra --no-cpu-baseline --no-end-to-end $v > gpurun_out/r03/r03t_c3.json 2>/dev/null
  python - "$v" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r03/r03t_c3.json').read().strip().splitlines()[-1])
print('c3 %-16s %.4g rays/s  %.3f ms per step' % (sys.argv[1] or 'rows + histogram', d['value'], d['ms_per_step']))
PY
done
