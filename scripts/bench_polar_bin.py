#!/usr/bin/env python3
"""the polar binning of the sweep's measure by itself: 1e7 rows of GettingStarted, calcFwhm's histogram ten times
(run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration, nothing else on the GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
pr = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
n = 10_000_000
with Tracer(0) as tr:
  tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
  tr.reserveHits(n + n // 8)
  tr.reset(); tr.trace(0, n, 5); tr.sync()
  dh = tr.deviceHits(None)
  H = dh.histogram(**sweep._FWHM_BINS)
  t0 = time.perf_counter()
  for _ in range(10):
    H = dh.histogram(**sweep._FWHM_BINS)
  dt = (time.perf_counter() - t0) / 10
  print(f'{len(dh)} rows, {H.hist.sum():.0f} binned, {dt * 1e3:.3f} ms per histogram() (plane search, projection, medians and binning)')
