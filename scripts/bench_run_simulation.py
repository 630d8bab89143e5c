#!/usr/bin/env python3
"""whole-product throughput: runSimulation (continuous mode, hit rows into the run folder / into memory) on
lensesAndMirrors, EndAfterRays = n.   python scripts/bench_run_simulation.py [n] [raysPerLaunch]"""
import cProfile, json, os, pstats, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import simulation_loop
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8
rpl = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 22
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd'))
doc.OpticalSimulationSettings.EndAfterRays = '%g' % n
for where in ('memory', 'disk'):
  tmp = tempfile.mkdtemp(prefix='odw_run_', dir='/tmp') if where == 'disk' else None
  for rep in range(2):
    t0 = time.perf_counter()
    pr = cProfile.Profile() if rep == 1 else None
    if pr: pr.enable()
    store = simulation_loop.runSimulation(doc, 'true', resultsPath=tmp, raysPerLaunch=rpl, compileScene='structure')
    if pr: pr.disable()
    dt = time.perf_counter() - t0
  print(json.dumps(dict(where=where, rays=store.totalTracedRays, hits=store.totalRecordedHits, seconds=round(dt, 3),
                        rays_per_s=float('%.3g' % (store.totalTracedRays / dt)), rays_per_launch=rpl)), flush=True)
  pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
  if tmp:
    shutil.rmtree(tmp, ignore_errors=True)
