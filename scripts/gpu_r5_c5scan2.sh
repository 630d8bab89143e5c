#!/bin/bash
# round 5: after the shared ray pass -- contexts in flight, blocks per CU of the batch launches, group size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_c5_scan2.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O || exit 1
run() {
  echo "== $*" | tee -a $O
  env "$@" timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('avg_kernel_ms',) if k in d['roofline']})" | tee -a $O
}
for rep in 1 2; do
  run X=1 && run ODW_SWEEP_PIPELINE=2 && run ODW_SWEEP_PIPELINE=4 && run ODW_BATCH_GRID_MULT=4 && run ODW_BATCH_GRID_MULT=2 && run ODW_SWEEP_BATCH=8 && run ODW_SWEEP_BATCH=24 || exit 1
done
