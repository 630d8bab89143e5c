#!/usr/bin/env python3
"""the radius sweep of C5, sweep by sweep: wall time of each of N sweeps (bench.py prints their mean), with the garbage
collector as it is and switched off for the sweeps
  python scripts/c5_variance.py [n_sweeps]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
radii = np.linspace(9, 11, 64)
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))


def setRadius(d, r):
  d.Sphere.Radius = float(r)


tr = Tracer(0)
setRadius(doc, radii[0])
first = scenes.bakeProject(doc)
tr.setScene(first.scene)
tr.setLimits(first.limits)
tr.compileScene('structure')


def run():
  tr.sync()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  sweep.parameterSweep(doc, setRadius, radii, rays=10_000_000, seed=0x0D15EA5E, tracer=tr,
                       measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot), keepSample=1000)
  tr.sync()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) * 1e3


run()
for mode in ('gc on', 'gc off', 'gc on', 'gc off'):
  if mode == 'gc off':
    gc.collect()
    gc.disable()
  else:
    gc.enable()
  ms = [run() for _ in range(n)]
  print(mode, 'ms per sweep:', ' '.join(f'{v:.1f}' for v in ms), '| median', f'{np.median(ms):.1f}', 'max', f'{max(ms):.1f}', flush=True)
gc.enable()
