#!/bin/bash
# repeatability of the compiled kernel's time: the same command several times in one call
cd "$(dirname "$0")/.."
for k in 1 2 3 4; do
  timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --compile structure "$@" > gpurun_out/spec_rep.log 2>&1 || { echo failed; tail -5 gpurun_out/spec_rep.log; exit 1; }
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/spec_rep.log').read().strip().splitlines()[-1])
print('%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'], d['config']['scene_compiled'], flush=True)
PY
done
