#!/usr/bin/env python3
"""random crowded scenes (17 - 64 analytic primitives scattered in space, tests/random_scenes.py): kernel ms of
4e6 explicit rays through whatever kernel the library picks -- run with ODW_BVH_THRESHOLD / ODW_SPEC_MAX_PRIMS /
ODW_COMPILE set to compare the grid kernel, the generic flat kernel and compiled kernels beyond 16 primitives"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from random_scenes import rays, scene
n = 4_000_000
picked = []
for s in range(200):
  rs = np.random.RandomState(7000 + s)
  try:
    sc, lim, targets = scene(rs, crowded=True)
  except Exception:
    continue
  k = len(sc.prim_type)
  if 17 <= k <= 64 and not any(int(t) >= 5 for t in sc.prim_type):
    picked.append((k, sc, lim, targets, rs))
  if len(picked) == 6:
    break
for k, sc, lim, targets, rs in sorted(picked, key=lambda x: x[0]):
  o, d = rays(rs, targets, n)
  with Tracer(0) as tr:
    tr.setScene(sc); tr.setLimits(lim); tr.setDetector(None)
    tr.reserveHits(8 * n)
    tr.timingEnable(True)
    best = 1e9
    for _ in range(3):
      tr.reset(); tr.timingRead()
      tr.traceRays(o, d)
      tr.sync()
      best = min(best, tr.timingRead()[0])
    c = tr.counters()
    print(json.dumps(dict(prims=k, ms=round(best, 3), compiled=tr.compiledInfo()['mode'], segments_per_ray=round(c['segments'] / n, 2),
                          hits_per_ray=round(c['recorded_hits'] / n, 2))), flush=True)
