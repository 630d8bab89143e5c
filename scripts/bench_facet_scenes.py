"""Kernel time of the reference's scenes that reach the tracer as facets: the housing of
imported-stepfile-as-surface-source.FCStd (4e4 facets, a surface source) and the cemented achromat of
edmund-optics-lens.FCStd with the exact-CSG recognition switched off (1.2e4 facets).  ODW_MESH_KERNEL=0: the binary
BVH kernel instead of the eight-wide one."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import geometry
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for name, exact in (('imported-stepfile-as-surface-source', True), ('edmund-optics-lens', False), ('edmund-optics-lens', True)):
  geometry.BREP_EXACT = exact
  pr = scenes.bakeProject(f'tests/golden/scenes/{name}.FCStd')
  ntri = int((pr.scene.prim_type == geometry.TRIANGLE).sum())
  with Tracer(0) as tr:
    tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
    tr.reserveHits(int(n * 2.5) + 1024)
    tr.reset(); tr.trace(1 << 40, n, 1, histogram=False); tr.sync()
    tr.timingEnable(True); tr.timingRead()
    for s in range(3):
      tr.reset(); tr.trace(s * n, n, 1, histogram=False)
    tr.sync()
    ms, k = tr.timingRead()
    c = tr.counters()
    print('%-40s facets %6d  kernel %.3f ms -> %.3g rays/s (%.2f segments per ray)' %
          (name + ('' if exact else ' (facets)'), ntri, ms / k, n * k / ms * 1e3, c['segments'] / c['traced_rays']), flush=True)
