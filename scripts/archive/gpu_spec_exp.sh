#!/bin/bash
# A/B of the scene-compiled flat kernels on C3 (ODW_COMPILE picked up by every Tracer)
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/spec_exp.log 2>&1 || { echo "failed"; tail -5 gpurun_out/spec_exp.log; return 1; }
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/spec_exp.log').read().strip().splitlines()[-1])
print('%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'], d['config']['segments_per_ray'], d['config']['hits_per_ray'], flush=True)
PY
}
echo "generic"; run --compile off "$@" &&
echo "structure" && run --compile structure "$@"
