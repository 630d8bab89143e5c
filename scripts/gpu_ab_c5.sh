#!/bin/bash
# A/B of library builds on C5 (bench.py --config c5, 5 sweeps after one warm-up sweep): the in-tree library against
# build/libodw_c5_*.so, twice each, alternating
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for lib in freecad/optics_design_workbench_amd/csrc/libodw_trace.so build/libodw_c5_*.so; do
    ODW_TRACE_LIB=$PWD/$lib timeout -k 10 200 python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],2), 'ms/sweep', '%.3g rays/s' % d['value'])"
  done
done
