#!/usr/bin/env python3
"""kernel time of consecutive C3 launches (HIP events), generic and compiled: warm-up behaviour"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
pr = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd'))
det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=1024, ny=1024, toward=pr.source.xform[[3, 7, 11]])
for mode in ('off', 'structure'):
  tr = Tracer(0)
  tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(det)
  tr.compileScene(mode)
  tr.reserveHits(n + 1024)
  tr.timingEnable(True)
  times = []
  for k in range(14):
    tr.resetHits() if k else tr.reset()
    tr.timingRead()
    tr.trace(k * n, n, 0x0D15EA5E)
    tr.sync()
    times.append(round(tr.timingRead()[0], 2))
  print(mode, times, flush=True)
  tr.close()
