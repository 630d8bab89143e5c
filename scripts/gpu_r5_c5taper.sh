#!/bin/bash
# round 5: sizes of the sweep's last groups (ODW_SWEEP_TAPER)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_c5_taper.log
: > $O
for rep in 1 2; do
  for M in ${TAPERS:-0 2 3 4 6}; do
    echo "== ODW_SWEEP_TAPER=$M" | tee -a $O
    ODW_SWEEP_TAPER=$M timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d.get(k) for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('avg_kernel_ms',) if k in d['roofline']})" | tee -a $O || exit 1
  done
done
