#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c5
export TMPDIR=/tmp
d=gpurun_out/r05_c5/prof_tl
rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace -d $d -- python3 bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extra > $d.log 2>&1 || { echo failed; tail -5 $d.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*' $d.log | tail -1
python3 scripts/ktimeline.py $d --schema
python3 scripts/ktimeline.py $d ${1:-110} | tee gpurun_out/r05_c5/timeline_${2:-a}.log
rm -rf $d
