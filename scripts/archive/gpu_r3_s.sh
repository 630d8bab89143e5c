#!/bin/bash
# round 3: the default bench line with the sampled shader clock
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/r03s_default_bench.json 2> gpurun_out/r03/r03s_err.log || { tail -20 gpurun_out/r03/r03s_err.log; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/r03s_default_bench.json').read().strip().splitlines()[-1])
print('c3', '%.4g' % d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('frac_at_sampled_clock'), d.get('clock'))
for k,v in d.get('extra_configs',{}).items(): print(k, '%.4g' % v.get('value',0), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('frac_at_sampled_clock'), v.get('clock'))
PY
