import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
proj = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd'))
tr = Tracer(0)
tr.setScene(proj.scene); tr.setSource(proj.source); tr.setLimits(proj.limits); tr.setDetector(None)
tr.reserveHits(n + 1024)
for rep in range(2):
  tr.reset()
  t0 = time.perf_counter(); tr.trace(0, n, 0x0D15EA5E); tr.sync(); t1 = time.perf_counter()
  h = tr.hits(); t2 = time.perf_counter()
  ray = (h['tag'] & np.uint64(0xFFFFFFFFFFFF))
  assert np.all(np.diff(ray.astype(np.int64)) >= 0)
  print(json.dumps(dict(rays=n, hits=len(h), trace_s=t1 - t0, fetch_sorted_s=t2 - t1, fetch_GBps=len(h) * 64 / (t2 - t1) / 1e9,
                        end_to_end_rays_per_s=n / (t2 - t0))))
