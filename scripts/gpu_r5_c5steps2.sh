#!/bin/bash
# round 5: kernels of the batched chain alone, for the default library and variant builds
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c5
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_device_hits.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for V in default "$@"; do
  if [ $V = default ]; then unset ODW_TRACE_LIB; else export ODW_TRACE_LIB=build/libodw_$V.so; fi
  d=gpurun_out/r05_c5/prof_steps
  rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -- python3 scripts/measure_batch_steps.py 1e7 8 > $d.log 2>&1 || { echo failed; tail -5 $d.log; exit 1; }
  python3 scripts/kstats.py $d 12 > gpurun_out/r05_c5/kernels_$V.csv
  rm -rf $d
  echo "== $V"
  python3 - <<'PY' gpurun_out/r05_c5/kernels_$V.csv
import sys
for l in open(sys.argv[1]).read().splitlines()[1:]:
    p=l.rsplit(',',4); print(f"{p[0][:70]:70s} calls {p[1]:>5} total_us {float(p[2]):11.1f} avg_us {float(p[3]):9.1f}")
PY
done
