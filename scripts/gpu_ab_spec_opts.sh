#!/bin/bash
# A/B of compiler switches of the scene-compiled kernel (ODW_SPEC_OPTS reaches hiprtc) on bench.py --config CONFIG:
#   bash scripts/gpu_ab_spec_opts.sh c3 "-DODW_FLAT_NOBRANCH=0" "-DODW_FLAT_NOBRANCH=15" ...     (twice each, alternating)
cd "$GRAFT_REPO_ROOT"
cfg=$1; shift
for round in 1 2; do
  for o in "$@"; do
    ODW_SPEC_OPTS="$o" timeout -k 10 200 python bench.py --config $cfg --no-extra --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$o', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['avg_kernel_ms'],3), 'kernel ms', d['roofline']['kernel'])"
  done
done
