#!/bin/bash
# round 5: where the time of a sorted grid launch goes (key pass, radix sort, trace kernel), rocprofv3 kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c4sorted
export TMPDIR=/tmp
for c in 0:1:1 12:1:1 16:1:1 20:1:1 32:1:1; do
  d=gpurun_out/r05_c4sorted/prof_${c//:/_}
  rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $d -- python3 scripts/ablate_c4_sorted.py 1.25e8 $c > $d.log 2>&1 || { echo failed $c; tail -5 $d.log; exit 1; }
  echo "== $c"
  python3 scripts/kstats.py $d 8 | tee $d.csv
  rm -rf $d
done
