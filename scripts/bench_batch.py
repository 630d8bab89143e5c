#!/usr/bin/env python3
"""kernel time of a batch launch against the launches it replaces (GettingStarted, compiled and generic kernels)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prs = []
for r in np.linspace(9, 11, S):
  doc.Sphere.Radius = float(r); prs.append(scenes.bakeProject(doc))
for mode in ('structure', 'off'):
  with Tracer(0) as tr:
    tr.compileScene(mode)
    tr.setLimits(prs[0].limits); tr.setSource(prs[0].source); tr.setDetector(None)
    cap = int(n * 1.25) + 1024
    def timed(f, reps=5):
      f(); tr.sync()
      tr.timingEnable(True); tr.timingRead()
      t = time.perf_counter()
      for _ in range(reps): f()
      tr.sync(); w = (time.perf_counter() - t) / reps
      ms, launches = tr.timingRead()
      return ms / reps, w * 1e3
    def singles():
      for pr in prs:
        tr.setScene(pr.scene); tr.reserveHits(cap); tr.reset(); tr.trace(0, n, 7, histogram=False)
    k, w = timed(singles)
    print(f'{mode:9s} {S} single launches of {n:.0e}: kernel {k:8.3f} ms  wall {w:8.3f} ms')
    tr.setScene(prs[0].scene); tr.reserveHits(int(S * n * 1.25) + 1024)
    def big():
      tr.reset(); tr.trace(0, S * n, 7, histogram=False)
    k, w = timed(big)
    print(f'{mode:9s} one launch of {S * n:.0e} (one scene): kernel {k:8.3f} ms  wall {w:8.3f} ms')
    tr.reserveHits(16)
    tr.setSceneBatch([pr.scene for pr in prs])
    def batch():
      tr.reset(); tr.traceBatch(0, n, 7, cap)
    k, w = timed(batch)
    print(f'{mode:9s} batch launch {S} x {n:.0e}: kernel {k:8.3f} ms  wall {w:8.3f} ms', tr.counters()['hits_dropped'])
