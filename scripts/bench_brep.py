#!/usr/bin/env python3
"""Throughput of the reference's cemented achromat (edmund-optics-lens.FCStd, two STEP imports):
recognised as exact CSG (scene/brep_csg.py, the default), as facets (triangle primitives + BVH),
and built by hand from exact spheres and a cylinder.  Measured on one MI355X, 2e7 rays:
4.4e9 / 1.0e9 / 4.3e9 rays/s.
  python scripts/bench_brep.py [rays]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

from freecad.optics_design_workbench_amd.scene import bake, geometry
from freecad.optics_design_workbench_amd.simulation.simulation_loop import bakeLightSource
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

import test_brep

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
doc, ref = test_brep._achromat_documents()
with Tracer(0) as tr:
  for name, d, exact in (('achromat, BRep import recognised as exact CSG', doc, True),
                         ('achromat, BRep import as facets + BVH', doc, False),
                         ('achromat, built from exact spheres + cylinder', ref, True)):
    geometry.BREP_EXACT = exact
    src = bake.lightSources(d)[0]
    t0 = time.perf_counter()
    sc, bs, lim = bake.bakeScene(d, src), bakeLightSource(d, src, 0), bake.bakeLimits(d, src)
    t_bake = time.perf_counter() - t0
    tr.setScene(sc); tr.setSource(bs); tr.setLimits(lim); tr.setDetector(None)
    tr.reserveHits(n + 1024)
    for rep in range(2):
      tr.reset()
      tr.sync()
      t0 = time.perf_counter()
      tr.trace(0, n, 3, histogram=False)
      tr.sync()
      dt = time.perf_counter() - t0
    c = tr.counters()
    print(json.dumps(dict(config=name, rays=n, seconds=dt, rays_per_s=n / dt, prims=int(len(sc.prim_type)),
                          bake_s=round(t_bake, 2), segments_per_ray=c['segments'] / n, hits_per_ray=c['recorded_hits'] / n)))
