#!/bin/bash
# What each piece of the compiled flat kernel costs as it runs: the piece is computed twice (-DODW_DOUBLE=k through
# ODW_SPEC_OPTS, results unchanged), the launch's extra time against k = 0 is its cost.   bash scripts/gpu_double_profile.sh [c3|c5]
cd "$GRAFT_REPO_ROOT"
cfg=${1:-c3}
names=(baseline "box tests" "sphere roots" "cylinder side+caps (whole candidate pass)" "box faces (whole candidate pass)" "trimming tests" "normal at the hit" "mirror / Snell" "ray generation" "inverse direction" x x "sphere (whole candidate pass)")
for k in 0 1 2 12 3 4 5 6 7 8 9 0; do
  ODW_SPEC_OPTS="-DODW_DOUBLE=$k" timeout -k 10 200 python bench.py --config $cfg --steps 10 --warmup 2 --no-extra --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ODW_DOUBLE=$k', '%-44s' % '${names[$k]}', round(d['roofline']['avg_kernel_ms'],3), 'kernel ms')"
done
