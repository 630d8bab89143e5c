"""Launch length against what lies between launches (GettingStarted, compiled kernel): resets between the launches
or none (rows pile up in one list), 1e7 rays."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
pr = scenes.bakeProject('tests/golden/scenes/GettingStarted.FCStd')
tr = Tracer(0)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
tr.compileScene('structure')
n, reps = 10_000_000, 8
tr.reserveHits(int(n * 1.25 * reps) + 1024)
for mode in ('reset between', 'back to back', 'reset between', 'back to back', 'no rows', 'no rows, histogram'):
  tr.reset(); tr.trace(1 << 40, n, 1, histogram=False); tr.sync()
  tr.reset()
  tr.timingEnable(True); tr.timingRead()
  t0 = time.perf_counter()
  for s in range(reps):
    if mode == 'reset between':
      tr.reset()
    tr.trace(s * n, n, 1, histogram=(mode == 'no rows, histogram'), record_hits=not mode.startswith('no rows'))
  tr.sync()
  wall = (time.perf_counter() - t0) / reps * 1e3
  ms, k = tr.timingRead(); tr.timingEnable(False)
  print('%-22s kernel %.4f ms, wall %.4f ms per launch' % (mode, ms / k, wall), flush=True)
