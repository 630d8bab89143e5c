#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp ODW_PROFILE_ROUND=r04
T=${1:-r04a}
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -2
python scripts/profile_round.py ${T}_c5 --config c5 > $O/${T}_c5_profile.log 2>&1 || { tail -30 $O/${T}_c5_profile.log; exit 1; }
tail -2 $O/${T}_c5_profile.log | cut -c1-300
# the sweep as it runs by default (batch launches): kernel trace of two sweeps
rocprofv3 --kernel-trace --stats -d $O/${T}_c5_batch_trace -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/${T}_c5_batch_bench.json 2> /dev/null
python - $O/${T}_c5_batch_trace $O/${T}_c5_batch_kernel_stats.csv <<'PY'
import glob, sqlite3, sys
con = sqlite3.connect(sorted(glob.glob(sys.argv[1] + '/**/*.db', recursive=True))[-1])
name = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith('top_kernels')][0]
cols = [c[1] for c in con.execute(f'pragma table_info({name})')]
with open(sys.argv[2], 'w') as f:
  f.write(','.join(cols) + '\n')
  for r in con.execute(f'select * from {name}'):
    f.write(','.join(str(x) for x in r) + '\n')
PY
rm -rf $O/${T}_c5_batch_trace
head -8 $O/${T}_c5_batch_kernel_stats.csv | cut -c1-160
rm -f $O/${T}_*_pmc?.log $O/${T}_*_trace.log
