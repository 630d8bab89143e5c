#!/bin/bash
# round 5: after the polling wait -- contexts in flight, group size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_c5_scan4.log
: > $O
run() {
  echo "== $*" | tee -a $O
  env "$@" timeout -k 10 300 python bench.py --config c5 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('avg_kernel_ms',) if k in d['roofline']})" | tee -a $O
}
for rep in 1 2; do
  run X=1 && run ODW_SWEEP_PIPELINE=2 && run ODW_SWEEP_PIPELINE=4 && run ODW_SWEEP_PIPELINE=5 && run ODW_SWEEP_BATCH=8 && run ODW_SWEEP_BATCH=12 && run ODW_SWEEP_BATCH=20 && run ODW_BATCH_GRID_MULT=4 || exit 1
done
