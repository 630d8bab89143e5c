#!/bin/bash
# A/B of grid-kernel variants on hugeArray (1.25e8 rays): kernel ms by HIP events, hit rows + histogram
cd "$GRAFT_REPO_ROOT"
for lib in build/libodw_c4_*.so; do
  echo "== $lib"
  ODW_TRACE_LIB=$PWD/$lib python scripts/ablate_c4.py 1.25e8 2>&1 | head -1
done
