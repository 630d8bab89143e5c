#!/bin/bash
# round 3, call E: isolated-solid shortcut: parity suites + A/B on c3 / c5
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_geometry.py tests/test_gpu_compiled.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r03e_tests.log 2>&1 || { tail -40 gpurun_out/r03e_tests.log; exit 1; }
tail -2 gpurun_out/r03e_tests.log
for iso in 1 0; do
  ODW_ISOLATED=$iso python bench.py --no-cpu-baseline --no-end-to-end --no-extra --steps 10 > gpurun_out/r03e_c3_iso$iso.json 2>gpurun_out/r03e_err.log || { tail -20 gpurun_out/r03e_err.log; exit 1; }
  ODW_ISOLATED=$iso python bench.py --no-cpu-baseline --no-end-to-end --no-extra --steps 10 --compile off > gpurun_out/r03e_c3gen_iso$iso.json 2>gpurun_out/r03e_err.log || { tail -20 gpurun_out/r03e_err.log; exit 1; }
  python - $iso <<'PY'
import json,sys
for tag in ('c3','c3gen'):
  d=json.loads(open(f'gpurun_out/r03e_{tag}_iso{sys.argv[1]}.json').read().strip().splitlines()[-1])
  print('isolated', sys.argv[1], tag, '%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'])
PY
done
