#!/bin/bash
# round 4: smoke, the whole GPU suite, the default bench line, kernel trace + counter passes for c3 / c4 / c5
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp ODW_PROFILE_ROUND=r04
T=${1:-r04a}
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_smoke.log 2>&1 || { tail -30 $O/${T}_smoke.log; exit 1; }
tail -1 $O/${T}_smoke.log
if [ "$2" != "notests" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1 || { tail -60 $O/${T}_gpu_tests.log; exit 1; }
  tail -1 $O/${T}_gpu_tests.log
fi
python bench.py > $O/${T}_default_bench.json 2>$O/${T}_default_bench.err || { tail -20 $O/${T}_default_bench.err; exit 1; }
cp $O/bench_detail.json $O/${T}_default_bench_detail.json
for c in c3 c4 c5; do
  python scripts/profile_round.py ${T}_$c --config $c > $O/${T}_${c}_profile.log 2>&1 || { tail -30 $O/${T}_${c}_profile.log; exit 1; }
  tail -2 $O/${T}_${c}_profile.log | cut -c1-300
done
rm -f $O/${T}_*_pmc?.log $O/${T}_*_trace.log
