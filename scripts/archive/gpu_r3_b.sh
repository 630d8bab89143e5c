#!/bin/bash
# round 3, call B: post-hoc selection medians + pipelined sweep: tests, c5 line, kernel stats of the sweep
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_device_hits.py tests/test_gpu_scale.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r03b_tests.log 2>&1 || { tail -40 gpurun_out/r03b_tests.log; exit 1; }
tail -3 gpurun_out/r03b_tests.log
python bench.py --config c5 --warmup 1 --no-cpu-baseline > gpurun_out/r03b_c5.json 2> gpurun_out/r03b_c5.err || { tail -30 gpurun_out/r03b_c5.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b_c5.json').read().strip().splitlines()[-1])
print('c5', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/r03b_c5_trace -o c5 --output-format csv -- python3 bench.py --config c5 --warmup 1 --no-cpu-baseline > gpurun_out/r03b_c5_trace.log 2>&1 || { tail -20 gpurun_out/r03b_c5_trace.log; exit 1; }
find gpurun_out/r03b_c5_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03b_c5_kernel_stats.csv
rm -rf gpurun_out/r03b_c5_trace
cut -c1-60,400- gpurun_out/r03b_c5_kernel_stats.csv | head -5
