#!/bin/bash
# round 3, call O: mesh kernel with stochastic surfaces -- parity; grid walk threshold after the ring-fill change
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_mesh.py tests/test_brep.py tests/test_gpu_fuzz.py tests/test_gpu_parity_geometry.py -m gpu -x -q > $O/r03o_tests.log 2>&1 || { tail -60 $O/r03o_tests.log; exit 1; }
tail -2 $O/r03o_tests.log
timeout -k 10 800 python tests/fuzz_parity.py 150 10000 621 1 > $O/r03o_fuzz_rich.log 2>&1 || true
tail -1 $O/r03o_fuzz_rich.log
for lib in "" build/libodw_gstep16.so build/libodw_gstep32.so; do
  ODW_TRACE_LIB=${lib:+$PWD/$lib} python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline > $O/r03o_c4.json 2>$O/r03o_err.log || { tail -20 $O/r03o_err.log; exit 1; }
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03/r03o_c4.json').read().strip().splitlines()[-1])
print('c4', sys.argv[1] or 'default', '%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'])
PY
done
