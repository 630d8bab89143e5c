#!/usr/bin/env python3
"""generic against compiled flat kernel on random scenes of growing primitive count (up to the flat
kernel's limit of 16): compile time, code size effects, rays/s.  python scripts/bench_compiled_sizes.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from random_scenes import rays, scene

n = 4_000_000
want = {4: None, 8: None, 12: None, 16: None}
for s in range(400):
  rs = np.random.RandomState(9000 + s)
  try:
    sc, lim, targets = scene(rs, crowded=(s % 2 == 1))
  except Exception:
    continue
  k = len(sc.prim_type)
  for w in want:
    if want[w] is None and w - 1 <= k <= w and not any(int(t) >= 5 for t in sc.prim_type):
      want[w] = (sc, lim, targets, rs)
  if all(v is not None for v in want.values()):
    break
for w, v in want.items():
  if v is None:
    continue
  sc, lim, targets, rs = v
  o, d = rays(rs, targets, n)
  out = dict(prims=len(sc.prim_type), types=np.bincount(np.asarray(sc.prim_type), minlength=5).tolist())
  for mode in ('off', 'structure'):
    with Tracer(0) as tr:
      tr.setScene(sc); tr.setLimits(lim); tr.setDetector(None)
      t0 = time.perf_counter()
      info = tr.compileScene(mode)
      out[mode + '_bind_s'] = round(time.perf_counter() - t0, 2)
      tr.reserveHits(8 * n)
      tr.timingEnable(True)
      best = 1e9
      for _ in range(3):
        tr.reset(); tr.timingRead()
        tr.traceRays(o, d)
        tr.sync()
        best = min(best, tr.timingRead()[0])
      c = tr.counters()
      out[mode + '_ms'] = round(best, 3)
      out['segments_per_ray'] = round(c['segments'] / n, 2)
      out[mode + '_mode'] = info['mode']
  print(json.dumps(out), flush=True)
