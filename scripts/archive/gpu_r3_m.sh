#!/bin/bash
# round 3, call M: histogram window centred on the mean of the first hits -- c3 with / without histogram, counters
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python bench.py --no-cpu-baseline --no-end-to-end --no-extra --steps 10 > $O/r03m_c3.json 2>$O/r03m_err.log || { tail -20 $O/r03m_err.log; exit 1; }
python bench.py --no-cpu-baseline --no-end-to-end --no-extra --steps 10 --no-histogram > $O/r03m_c3_nohist.json 2>$O/r03m_err.log || { tail -20 $O/r03m_err.log; exit 1; }
python bench.py --no-cpu-baseline --no-end-to-end --no-extra --steps 10 --compile off > $O/r03m_c3_generic.json 2>$O/r03m_err.log || { tail -20 $O/r03m_err.log; exit 1; }
python - <<'PY'
import json
for t in ('c3','c3_nohist','c3_generic'):
  d=json.loads(open(f'gpurun_out/r03/r03m_{t}.json').read().strip().splitlines()[-1])
  print(t, '%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'])
PY
python scripts/profile_round.py r03m_c3 --config c3 > $O/r03m_prof.log 2>&1 || { tail -30 $O/r03m_prof.log; exit 1; }
cp gpurun_out/r03m_c3_pmc.json gpurun_out/r03m_c3_kernel_stats.csv gpurun_out/r03m_c3_pmc_current.json gpurun_out/r03m_c3_bench.json $O/ 2>/dev/null || true
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03m_c3_pmc.json'))
print('write GB', d['write_bytes']/1e9, 'fetch', d['fetch_bytes_corrected']/1e9, 'ms', d['kernel_ms_rocprof'], 'valu', d['counters_avg_per_dispatch'].get('SQ_INSTS_VALU'))
PY
