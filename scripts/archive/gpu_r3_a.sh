#!/bin/bash
# round 3, call A: multi tests, default bench line (c3 + extra_configs), profiles of c3 / c4 / c5
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r03a_multi.log 2>&1 || { tail -30 gpurun_out/r03a_multi.log; exit 1; }
tail -3 gpurun_out/r03a_multi.log
python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err || { tail -30 gpurun_out/r03a_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03a_bench.json').read().strip().splitlines()[-1])
print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('wavefront_equivalent_frac'))
for k,v in d.get('extra_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('error'))
PY
for c in c3 c4 c5; do
  python scripts/profile_round.py r03a_$c --config $c > gpurun_out/r03a_prof_$c.log 2>&1 || { tail -30 gpurun_out/r03a_prof_$c.log; exit 1; }
  tail -2 gpurun_out/r03a_prof_$c.log
done
