#!/bin/bash
# A/B of library builds on C3 (bench.py --config c3, 20 steps): build/libodw_c3_*.so, twice each, alternating
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for lib in build/libodw_c3_*.so; do
    ODW_TRACE_LIB=$PWD/$lib python bench.py --config c3 --no-extra --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['avg_kernel_ms'],3), 'kernel ms', d['roofline']['kernel'])"
  done
done
