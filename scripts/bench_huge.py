import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd import _native, scenes
if os.environ.get('ODW_VARIANT_LIB'):
  _native.LIB_PATH = os.path.abspath(os.environ['ODW_VARIANT_LIB'])
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25000000
proj = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'hugeArray.FCStd'))
tr = Tracer(0)
tr.setScene(proj.scene); tr.setSource(proj.source); tr.setLimits(proj.limits); tr.setDetector(None)
tr.reserveHits(int(os.environ.get('ODW_BENCH_HITS', n)))     # < 4M rows: one atomic per append; else block reservations
tr.trace(1 << 40, n, 1); tr.sync(); tr.reset()
tr.timingEnable(True); tr.timingRead()
t0 = time.perf_counter()
tr.trace(0, n, 0x0D15EA5E); tr.sync()
dt = time.perf_counter() - t0
ms, k = tr.timingRead()
c = tr.counters()
print(json.dumps(dict(rays=n, rays_per_s=n / dt, kernel_ms=ms, seg_per_ray=c['segments'] / n, hits=c['recorded_hits'], capped=c['capped'])))
