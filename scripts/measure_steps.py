#!/usr/bin/env python3
"""Where a sweep value's measure spends its time (one thread, nothing else running): the steps of DeviceHits behind
calcFwhm + rmsSpot on a 1e7-ray launch of GettingStarted, each timed with the stream drained before and after."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from freecad.optics_design_workbench_amd.simulation.device_hits import DeviceHits

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
tr = Tracer(0)
tr.compileScene('structure')
acc = {}
def T(name, f):
  tr.sync(); t = time.perf_counter(); r = f(); tr.sync(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
reps = 8
for rep in range(reps + 1):
  if rep == 1: acc.clear()
  doc.Sphere.Radius = 9.0 + 0.2 * rep
  pr = T('bake', lambda: scenes.bakeProject(doc))
  def up():
    tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None); tr.reserveHits(int(n * 1.25) + 1024); tr.reset()
  T('upload', up)
  T('trace', lambda: tr.trace(0, n, 7, histogram=False))
  T('counters', lambda: tr.counters())
  h = T('select', lambda: DeviceHits(tr))
  smp = T('sample (2 gathers)', lambda: h._sample())
  pn = T('plane search', lambda: h._flattest_direction(smp[0], 1e-9))
  T('calcFwhm total', lambda: sweep.calcFwhm(h))
  T('rms', lambda: sweep.rmsSpot(h))
  import cProfile, pstats
  if rep == reps:
    cProfile.run('sweep.calcFwhm(h)', '/tmp/prof_fwhm')
    pstats.Stats('/tmp/prof_fwhm').sort_stats('tottime').print_stats(25)
for k, v in acc.items():
  print(f'{k:22s} {1e3 * v / reps:8.3f} ms')
