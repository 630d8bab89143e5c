#!/bin/bash
# round 5: fits deferred + next group baked ahead (ODW_SWEEP_DEFER) against the order of round 5 so far
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_c5_defer.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_device_hits.py -m gpu -x -q -k "sweep or batch" 2>&1 | tail -3 | tee -a $O || exit 1
for rep in 1 2 3; do
  for M in 0 1; do
    echo "== ODW_SWEEP_DEFER=$M" | tee -a $O
    ODW_SWEEP_DEFER=$M timeout -k 10 300 python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('avg_kernel_ms',) if k in d['roofline']})" | tee -a $O || exit 1
  done
done
