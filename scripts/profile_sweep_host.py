#!/usr/bin/env python3
"""Where the host time of the C5 sweep goes: the sweep of bench.py --config c5, single-threaded (pipeline off) under
cProfile, then the wall time of the pipelined sweep for comparison.
  python scripts/profile_sweep_host.py [radii] [rays]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

argv = [a for a in sys.argv[1:] if not a.startswith('--')]
n_radii = int(argv[0]) if len(argv) > 0 else 64
rays = int(float(argv[1])) if len(argv) > 1 else 10_000_000
radii = np.linspace(9, 11, n_radii)
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))


def setRadius(d, r):
  d.Sphere.Radius = float(r)


tr = Tracer(0)
setRadius(doc, radii[0])
first = scenes.bakeProject(doc)
tr.setScene(first.scene)
tr.setLimits(first.limits)
tr.compileScene('structure')


def run(pipeline):
  return sweep.parameterSweep(doc, setRadius, radii, rays=rays, seed=1, tracer=tr, pipeline=pipeline,
                              measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot))


run(True)
for p in (True, 2, True, 2, True, 2, True, 2, True, 2, 3, 3, 3):
  t = time.perf_counter()
  run(p)
  print('pipeline', p, '%.1f ms per sweep' % (1e3 * (time.perf_counter() - t)), flush=True)
if '--profile' in sys.argv:
  pr = cProfile.Profile()
  pr.enable()
  run(False)
  pr.disable()
  st = pstats.Stats(pr)
  st.sort_stats('cumulative').print_stats(45)
