#!/bin/bash
# round 5: the batched post-hoc chain of the C5 sweep, step by step and kernel by kernel, nothing else on the GPU
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c5
export TMPDIR=/tmp
python3 scripts/measure_batch_steps.py 1e7 8 > gpurun_out/r05_c5/steps_${1:-a}.log 2>&1 || { tail -5 gpurun_out/r05_c5/steps_${1:-a}.log; exit 1; }
cat gpurun_out/r05_c5/steps_${1:-a}.log
d=gpurun_out/r05_c5/prof_steps
rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -- python3 scripts/measure_batch_steps.py 1e7 8 > $d.log 2>&1 || { echo failed; tail -5 $d.log; exit 1; }
python3 scripts/kstats.py $d 24 > gpurun_out/r05_c5/kernels_${1:-a}.csv
rm -rf $d
python3 - <<'PY' gpurun_out/r05_c5/kernels_${1:-a}.csv
import sys
for l in open(sys.argv[1]).read().splitlines()[1:]:
    p=l.rsplit(',',4); print(f"{p[0][:70]:70s} calls {p[1]:>5} total_us {float(p[2]):11.1f} avg_us {float(p[3]):9.1f}")
PY
