"""C5 sweep: what the wall time would be with a per-radius bake that costs nothing (memoised scenes), to tell how
much of the sweep is the interpreter lock."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd, bake as _bake
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
radii = np.linspace(9, 11, 64)
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
def setRadius(d, r):
  d.Sphere.Radius = float(r)
tr = Tracer(0)
setRadius(doc, radii[0])
first = scenes.bakeProject(doc)
tr.setScene(first.scene); tr.setLimits(first.limits); tr.compileScene('structure')
def run():
  return sweep.parameterSweep(doc, setRadius, radii, rays=10_000_000, seed=1, tracer=tr,
                              measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot))
run()
for _ in range(4):
  t = time.perf_counter(); run(); print('as is            %.1f ms per sweep' % (1e3 * (time.perf_counter() - t)), flush=True)
real = _bake.bakeScene
memo = {}
def cached(d, src=None, **kw):
  k = float(d.Sphere.Radius)
  if k not in memo:
    memo[k] = real(d, src, **kw)
  return memo[k]
_bake.bakeScene = cached
run()
for _ in range(4):
  t = time.perf_counter(); run(); print('bake memoised    %.1f ms per sweep' % (1e3 * (time.perf_counter() - t)), flush=True)
