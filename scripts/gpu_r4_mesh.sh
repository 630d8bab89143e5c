#!/bin/bash
# A/B of mesh-kernel builds (build/libodw_mesh_*.so): ball lens of 6.5e4 / 1.05e6 facets, 1e7 rays, kernel ms by HIP events
cd "$GRAFT_REPO_ROOT"
for lib in build/libodw_mesh_*.so; do
  echo "== $lib"
  ODW_TRACE_LIB=$PWD/$lib python scripts/bench_mesh.py --segments 256 1024 --rays 1e7 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('  ', d['case'], round(d['kernel_ms'], 3), 'ms', '%.3g' % d['rays_per_s'])"
done
