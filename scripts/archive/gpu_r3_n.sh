#!/bin/bash
# round 3, call N: grid kernel -- first cell found at the ring fill: parity (hugeArray, grid fuzz), C4 rate, phase times
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_parity_geometry.py tests/test_gpu_scale.py -m gpu -x -q > $O/r03n_tests.log 2>&1 || { tail -60 $O/r03n_tests.log; exit 1; }
tail -2 $O/r03n_tests.log
python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline > $O/r03n_c4.json 2>$O/r03n_err.log || { tail -20 $O/r03n_err.log; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/r03n_c4.json').read().strip().splitlines()[-1])
print('c4', '%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'])
PY
ODW_GRID_STATS=1 ODW_TRACE_LIB=$PWD/build/libodw_gstats.so python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "odw grid"
