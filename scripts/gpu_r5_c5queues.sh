#!/bin/bash
# round 5: C5 sweep against the number of hardware queues HIP spreads its streams over, and the trace grid's blocks per CU
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c5
for q in 4 8 16; do for m in 8 3; do
  GPU_MAX_HW_QUEUES=$q ODW_GRID_MULT=$m timeout -k 10 200 python3 bench.py --config c5 --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extra 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('hw_queues $q grid_mult $m  ms_per_sweep %.2f  avg_kernel_ms %s' % (d['ms_per_step'], d['roofline'].get('avg_kernel_ms')))
" | tee -a gpurun_out/r05_c5/queues_${1:-a}.log
done; done
