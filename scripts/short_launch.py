"""Kernel time of short and long launches of one scene (GettingStarted): where a launch's fixed cost goes.
   ODW_SL_COMPILE=off|structure, ODW_SL_REPS, ODW_CHUNK_RAYS / ODW_GRID_MULT are read by the library."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
pr = scenes.bakeProject('tests/golden/scenes/GettingStarted.FCStd')
tr = Tracer(0)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
tr.compileScene(os.environ.get('ODW_SL_COMPILE', 'structure'))
reps = int(os.environ.get('ODW_SL_REPS', '5'))
sizes = [int(float(x)) for x in os.environ.get('ODW_SL_SIZES', '1e7,1e8,1e6').split(',')]
for n in sizes:
  tr.reserveHits(int(n * 1.25) + 1024)
  tr.reset(); tr.trace(1 << 40, n, 1, histogram=False); tr.sync()
  tr.timingEnable(True); tr.timingRead()
  for s in range(reps):
    tr.reset(); tr.trace(s * n, n, 1, histogram=False)
  tr.sync()
  ms, k = tr.timingRead(); tr.timingEnable(False)
  print('rays %.0e x %d kernel %.4f ms -> %.3g rays/s  (%.4f ms per 1e6)' % (n, k, ms / k, n * k / ms * 1e3, ms / k / n * 1e6), flush=True)
