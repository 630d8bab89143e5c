#!/usr/bin/env python3
"""A train of k biconvex lenses (Common of two spheres and a cylinder: 3 primitives each) in front of a screen:
rays/s against the number of primitives, across the flat kernel's limit of 16 (generic / compiled flat
kernel up to 5 lenses, grid kernel beyond).  python scripts/bench_lens_train.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd.freecad_elements import make, point_source
from freecad.optics_design_workbench_amd.scene import Document, bake
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer


def scene(k):
  doc = Document()
  lenses = []
  for j in range(k):
    z = 30.0 + 12.0 * j
    a = make.makeSphere(doc, f'A{j}', 30.0, base=(0, 0, z + 28.0))       # front surface: centre behind the lens
    b = make.makeSphere(doc, f'B{j}', 30.0, base=(0, 0, z - 28.0))       # back surface
    c = make.makeCylinder(doc, f'C{j}', 6.0, 6.0, base=(0, 0, z - 3.0))
    lenses.append(make.makeCommon(doc, [a, b, c], f'L{j}'))
  make.makeLens(doc, lenses, RefractiveIndex=1.5)
  make.makeAbsorber(doc, [make.makeBox(doc, 'S', 60, 60, 1, base=(-30, -30, 30.0 + 12.0 * k + 20))], RecordHits=True)
  make.makeSimulationSettings(doc)
  src = make.makePointSource(doc, PowerDensity='exp(-theta**2/0.08**2)')
  return bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)


n = 20_000_000
for k in ((1, 3, 5, 6, 8, 12, 20) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
  sc, lim, src = scene(k)
  out = dict(lenses=k, prims=len(sc.prim_type))
  for mode in ('off', 'structure'):
    with Tracer(0) as tr:
      tr.setScene(sc); tr.setSource(src); tr.setLimits(lim); tr.setDetector(None)
      info = tr.compileScene(mode)
      tr.reserveHits(n + 1024)
      tr.timingEnable(True)
      best = 1e9
      for _ in range(3):
        tr.reset(); tr.timingRead()
        tr.trace(0, n, 1)
        tr.sync()
        best = min(best, tr.timingRead()[0])
      c = tr.counters()
      out[mode] = dict(ms=round(best, 3), rays_per_s=float('%.3g' % (n / best * 1e3)), compiled=info['mode'])
      out['segments_per_ray'] = round(c['segments'] / n, 2)
      out['hits_per_ray'] = round(c['recorded_hits'] / n, 3)
  print(json.dumps(out), flush=True)
