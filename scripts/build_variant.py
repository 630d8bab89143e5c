#!/usr/bin/env python3
"""another build of the HIP library with extra compiler flags, for A/B runs on the GPU box:
   python scripts/build_variant.py <name> [-DFLAG=VALUE ...]   ->  build/libodw_<name>.so
(picked up through ODW_TRACE_LIB, scripts/try_variants.sh runs bench.py with every build/libodw_*.so)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd import _native

name, flags = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, 'build'), exist_ok=True)
out = os.path.join(ROOT, 'build', f'libodw_{name}.so')
cmd = [_native.hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '-fPIC', '-shared'] + flags + \
      ['-o', out, os.path.join(_native.CSRC, 'odw_capi.hip')]
res = subprocess.run(cmd, cwd=_native.CSRC, capture_output=True, text=True)
if res.returncode:
  sys.exit(res.stdout + res.stderr)
print(out)
