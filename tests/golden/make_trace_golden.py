#!/usr/bin/env python3
"""Regression vectors of the tracing oracle (NOT reference outputs: the
reference's tracer needs FreeCAD/OpenCASCADE and cannot run here -- see
DESIGN.md section 3).  They freeze what the pinned-by-physics oracle produces
today, so that a later change of the oracle or the device shows up as a diff
against committed data:
    python tests/golden/make_trace_golden.py
writes tests/golden/trace_<scene>.npz: counters and the hit rows of rays
[first, first+n) of Philox stream `seed` for the shipped scene fixtures."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

CASES = {  # scene: (first, n, seed)
  'minimal': (0, 1500, 1), 'lensesAndMirrors': (10**6, 1500, 2), 'lensesAndMirrorsSequential': (7, 1500, 3),
  'GettingStarted': (0, 1500, 4), 'hugeArray': (0, 600, 5), 'grating': (0, 800, 6), 'mirror-diffuse': (0, 1500, 7),
  'simulation-modes-main': (0, 1500, 8),
}

if __name__ == '__main__':
  from conftest import project
  from oracle import capi
  for scene, (first, n, seed) in CASES.items():
    pr = project(scene)
    if hasattr(pr.source, 'face_prim'):
      r = capi.trace_surface(pr.scene, pr.source, pr.limits, first, n, seed)
    else:
      r = capi.trace(pr.scene, pr.source, pr.limits, first, n, seed)
    h = r['hits']
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', f'trace_{scene}.npz'),
                        first=first, n=n, seed=seed, counters=np.array([r['counters'][k] for k in capi.CNT_NAMES]),
                        point=h['point'], direction=h['direction'], power=h['power'], tag=h['tag'])
    print(scene, r['counters'])
