#!/bin/bash
# bench.py with every build/libodw_*.so in turn (variants compiled by hand with other flags or
# source experiments); prints rays/s and the kernel time of each.  Run on the GPU box.
# extra arguments go to bench.py (e.g. --config c4)
cd "$(dirname "$0")/.."
for lib in build/libodw_*.so; do
  ODW_TRACE_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/variant.log 2>&1 || { echo "$lib failed"; tail -3 gpurun_out/variant.log; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/variant.log').read().strip().splitlines()[-1])
print(sys.argv[1], '%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'], flush=True)
PY
done
