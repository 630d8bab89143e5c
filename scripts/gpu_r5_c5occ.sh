#!/bin/bash
# round 5: C5 sweep with the trace kernel's grid limited to k blocks per CU (room for post-hoc kernels of other groups beside it)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c5
for m in 8 4 3 2 8 3; do
  ODW_GRID_MULT=$m timeout -k 10 200 python3 bench.py --config c5 --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('grid_mult $m  ms_per_sweep %.2f  avg_kernel_ms %s' % (d['ms_per_step'], d['roofline'].get('avg_kernel_ms')))
" | tee -a gpurun_out/r05_c5/occ_${1:-a}.log
done
