#!/usr/bin/env python3
"""the steps of a batched measure (DeviceHitsBatch behind calcFwhm + rmsSpot), timed one by one on a batch of S scenes"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from freecad.optics_design_workbench_amd.simulation.device_hits import DeviceHitsBatch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr = Tracer(0); tr.compileScene('structure')
acc = {}
def T(name, f):
  tr.sync(); t = time.perf_counter(); r = f(); tr.sync(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
reps = 4
for rep in range(reps + 1):
  if rep == 1: acc.clear()
  prs = []
  def bake():
    for r in np.linspace(9 + .01 * rep, 11, S):
      doc.Sphere.Radius = float(r); prs.append(scenes.bakeProject(doc))
  T('bake', bake)
  def up():
    tr.setLimits(prs[0].limits); tr.setSource(prs[0].source); tr.setSceneBatch([p.scene for p in prs]); tr.setDetector(None); tr.reset()
  T('upload', up)
  T('trace', lambda: tr.traceBatch(0, n, 7, int(n * 1.25) + 1024))
  T('counters', lambda: tr.counters())
  b = T('select', lambda: DeviceHitsBatch(tr, S))
  T('sample+planes', lambda: b._detectPlanes())
  T('project+medians', lambda: b._project())
  H = T('bin', lambda: b.histograms(**sweep._FWHM_BINS))
  T('fits', lambda: [sweep._fwhmOfPolarHistogram(h) for h in H])
  T('rms', lambda: sweep.rmsSpot.batched(b))
for k, v in acc.items():
  print(f'{k:22s} {1e3 * v / reps:8.3f} ms per batch  {1e3 * v / reps / S:8.3f} per scene')
