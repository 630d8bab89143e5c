#!/bin/bash
# round 3, final pass: the whole GPU suite, the default bench line, counter passes for c3 / c4 / c5 and the mesh kernel
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
T=${1:-r03z}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_smoke.log 2>&1 || { tail -30 $O/${T}_smoke.log; exit 1; }
tail -1 $O/${T}_smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -60 $O/${T}_tests.log; exit 1; }
tail -1 $O/${T}_tests.log
python bench.py > $O/${T}_default_bench.json 2>$O/${T}_err.log || { tail -20 $O/${T}_err.log; exit 1; }
python - $T <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r03/{sys.argv[1]}_default_bench.json').read().strip().splitlines()[-1])
print('c3', '%.4g' % d['value'], d['ms_per_step'], d['roofline'].get('frac'), 'cpu', d.get('cpu_baseline',{}).get('value'))
for k,v in d.get('extra_configs',{}).items(): print(k, '%.4g' % v.get('value',0), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'))
PY
