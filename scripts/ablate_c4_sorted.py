#!/usr/bin/env python3
"""hugeArray (C4) launches with the grid kernel's rays handed out in sorted order (ODW_GRID_PRESORT = key bits) and the
wave's lanes interacting together (ODW_GRID_GATE = lanes), against the index order: kernel ms by HIP events (key pass
and sort included), and the results held against those of the index order (counters, histogram; rows at 1e7 rays)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 125000000
cases = [tuple(int(v) for v in c.split(':')) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else \
  [(0, 1, 1), (32, 1, 1), (32, 32, 1), (32, 48, 1), (32, 64, 1), (32, 64, 64), (24, 1, 1), (24, 48, 1), (24, 64, 1), (16, 1, 1), (16, 32, 1),
   (16, 48, 1), (16, 64, 1), (12, 48, 1), (0, 1, 1)]
pr = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'hugeArray.FCStd'))
gi = pr.scene.group_index('OpticalAbsorberGroup')
det = dict(group=gi, origin=[-0.5, -0.5, 61.0], ex=[1.0, 0.0, 0.0], ey=[0.0, 1.0, 0.0], x_lo=-25.0, x_hi=25.0, y_lo=-25.0, y_hi=25.0, nx=1024, ny=1024)
tr = Tracer(0)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(det)
tr.reserveHits(n // 2)
tr.timingEnable(True)
ref = None
for bits, gate, refill in cases:
  os.environ['ODW_GRID_PRESORT'] = str(bits)
  os.environ['ODW_GRID_GATE'] = str(gate)
  os.environ['ODW_GRID_REFILL'] = str(refill)
  best = 1e9
  for _ in range(3):
    tr.reset(); tr.timingRead()
    tr.trace(0, n, 0x0D15EA5E)
    tr.sync()
    best = min(best, tr.timingRead()[0])
  cnt, hist = tr.counters(), tr.histogram()
  same = None
  if ref is None:
    ref = (cnt, hist)
  else:
    same = cnt == ref[0] and bool(np.array_equal(hist, ref[1]))
  print(json.dumps(dict(presort_bits=bits, gate=gate, refill=refill, ms=round(best, 3), rays_per_s=round(n / best * 1e3, -6), same_results=same)), flush=True)
tr.close()
