#!/bin/bash
# round 5: the round's evidence on the library as committed -- GPU suite, smoke, the driver's bench line, profile passes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp ODW_PROFILE_ROUND=r05
T=${1:-r05a}
O=gpurun_out
step() { echo "== $*" | tee -a $O/${T}_final.log; }
: > $O/${T}_final.log
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  step "pytest -m gpu"
  timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee -a $O/${T}_final.log || exit 1
  step smoke
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/${T}_final.log || exit 1
fi
step "python bench.py"
timeout -k 10 600 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err || { tail -5 $O/${T}_bench_default.err; exit 1; }
cut -c1-1500 $O/${T}_bench_default.json | tee -a $O/${T}_final.log
for C in ${PROFILE_CONFIGS-c3 c4 c5}; do
  step "profile $C"
  timeout -k 10 900 python scripts/profile_round.py ${T}_$C --config $C > $O/${T}_${C}_profile.log 2>&1 || { tail -20 $O/${T}_${C}_profile.log; exit 1; }
  tail -3 $O/${T}_${C}_profile.log | cut -c1-300 | tee -a $O/${T}_final.log
  rm -f $O/${T}_${C}_pmc?.log $O/${T}_${C}_trace.log
done
