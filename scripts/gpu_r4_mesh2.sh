#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lib in build/libodw_mesh_*.so; do
  echo "== $lib"
  for sg in 0.05 0.12; do
  ODW_TRACE_LIB=$PWD/$lib python scripts/bench_mesh.py --segments 64 256 1024 --rays 1e7 --sigma $sg 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('  ', d['case'], 'sigma', d['sigma'], round(d['kernel_ms'], 3), 'ms', '%.3g' % d['rays_per_s'])"
  done
done
ODW_TRACE_LIB=$PWD/build/libodw_mesh_i60.so python scripts/bench_facet_scenes.py 2>&1 | tail -3
