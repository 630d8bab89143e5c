#!/bin/bash
# round 5: batch launches with the rays generated once for all scenes (ODW_BATCH_SHARED_RAYS) against every scene generating its own
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_c5_shared_rays.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O || exit 1
for rep in 1 2; do
  for M in 0 -1; do
    echo "== ODW_BATCH_SHARED_RAYS=$M" | tee -a $O
    if [ $M = -1 ]; then unset ODW_BATCH_SHARED_RAYS; else export ODW_BATCH_SHARED_RAYS=$M; fi
    timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('avg_kernel_ms', 'frac', 'pmc_stale') if k in d['roofline']})" | tee -a $O || exit 1
  done
done
