#!/bin/bash
# round 5: C3 against the size of the hit-list blocks a wave reserves per atomic (ODW_HIT_BLOCK): the slack of unused
# slots (WRITE_SIZE 8.19 GB for 6.40 GB of rows) against the number of atomics
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_c3_hit_block.log
: > $O
for B in 512 64 128 256 1024 2048 512; do
  echo "== ODW_HIT_BLOCK=$B" | tee -a $O
  ODW_HIT_BLOCK=$B timeout -k 10 300 python bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('avg_kernel_ms',) if k in d['roofline']})" | tee -a $O || exit 1
done
