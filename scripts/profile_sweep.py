#!/usr/bin/env python3
"""where the host time of the C5 radius sweep goes (cProfile over one sweep of 64 radii x 1e7 rays)"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
radii = np.linspace(9, 11, 64)
def setRadius(d, r):
  d.Sphere.Radius = float(r)
tr = Tracer(0)
tr.compileScene('structure')
run = lambda: sweep.parameterSweep(doc, setRadius, radii, rays=int(1e7), seed=1, tracer=tr, measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot))
run()
t0 = time.perf_counter(); run(); print('sweep seconds', time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
