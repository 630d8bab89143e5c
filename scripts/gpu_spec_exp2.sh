#!/bin/bash
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 150 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/spec_exp.log 2>&1 || { echo "failed"; tail -5 gpurun_out/spec_exp.log; return 1; }
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/spec_exp.log').read().strip().splitlines()[-1])
print('%.4g rays/s' % d['value'], '%.3f ms' % d['roofline']['avg_kernel_ms'], flush=True)
PY
}
export ODW_COMPILE=structure
for o in "-DODW_SPEC_WAVES=3" "-DODW_SPEC_WAVES=2" "-DODW_SPEC_WAVES=5" "-DODW_CHUNK=1024ull" "-DODW_REFILL_MIN=32" "-DODW_REFILL_MIN=8"; do
  echo "== $o"; ODW_SPEC_OPTS="$o" run || exit 1
done
