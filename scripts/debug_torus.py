import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_parity_geometry as T
from oracle import capi
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import Placement
np.set_printoptions(precision=15, linewidth=200)
for case in ('absorber_only', 'lens_only'):
  if case == 'absorber_only':
    sc, lim = T.build([('Absorber', lambda d: [make.makeTorus(d, 'T1', 10, 2, base=(0, 0, 0))], {})])
    ring = [[10 * np.cos(a), 10 * np.sin(a), 0] for a in np.linspace(0, 2 * np.pi, 40)]
  else:
    sc, lim = T.build([('Lens', lambda d: [make.makeTorus(d, 'T2', 6, 1.5, base=(0, 25, 3), quat=T.quat((1, 0, 0), 40))], dict(RefractiveIndex=1.5))])
    c = Placement(base=(0, 25, 3), quat=T.quat((1, 0, 0), 40))
    ring = [c * np.array([6 * np.cos(a), 6 * np.sin(a), 0]) for a in np.linspace(0, 2 * np.pi, 40)]
  o, d = T.aimed_rays(20000, ring, 1.2, 1)
  tr = Tracer(0)
  g, gc, r, rc = T.run_both(tr, capi, sc, lim, o, d)
  print(case, gc, rc)
  if len(g) == len(r) and np.array_equal(g['tag'], r['tag']):
    k = T.ordinal(r['tag'])
    dp = np.abs(g['point'] - r['point']).max(axis=1)
    for kk in range(0, 6):
      m = k == kk
      if m.any():
        print('  ordinal', kk, 'n', m.sum(), 'median %.2e p99 %.2e max %.2e' % (np.median(dp[m]), np.quantile(dp[m], .99), dp[m].max()))
    # residual of the torus equation on both sides (first hits, absorber case)
    if case == 'absorber_only':
      for name, h in (('gpu', g), ('oracle', r)):
        p = h['point']
        f = np.sqrt((np.sqrt(p[:, 0]**2 + p[:, 1]**2) - 10)**2 + p[:, 2]**2) - 2
        print('  ', name, 'surface residual median %.2e max %.2e' % (np.median(np.abs(f)), np.abs(f).max()))
      # and distance of hit from the ray line
      rid = (r['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
      for name, h in (('gpu', g), ('oracle', r)):
        v = h['point'] - o[rid]
        perp = v - (v * d[rid]).sum(1)[:, None] * d[rid]
        print('  ', name, 'off-line median %.2e max %.2e' % (np.median(np.linalg.norm(perp, axis=1)), np.linalg.norm(perp, axis=1).max()))
  tr.close()
