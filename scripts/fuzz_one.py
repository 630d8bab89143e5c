#!/usr/bin/env python3
"""one scene of tests/fuzz_parity.py on the mesh kernel, the binary kernels and the oracle (hits only, no segment rows:
the launch the parity run makes):  python scripts/fuzz_one.py seed scene rays rich"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from oracle import capi as oracle            # (a checker script, like tests/fuzz_parity.py)
from random_scenes import rays, scene
seed0, s, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rich, crowded, parab = sys.argv[4] in ('1', '2'), sys.argv[4] in ('2', '3', '5'), sys.argv[4] in ('4', '5')
rs = np.random.RandomState(seed0 * 100003 + s)
sc, lim, targets = scene(rs, rich, crowded, parab)
o, d = rays(rs, targets, n)
print('prims', np.bincount(sc.prim_type).tolist(), 'flags convex', int((sc.prim_flags & 2).astype(bool).sum()), 'groups', sc.group_type.tolist(),
      'seq', sc.seq_enabled, 'samplers', len(sc.surface_samplers), 'tol', lim.dist_tol, 'maxint', lim.max_intersections)
res = {}
for mode in ('1', '0'):
  os.environ['ODW_MESH_KERNEL'] = mode
  with Tracer(0) as tr:
    tr.setScene(sc); tr.setLimits(lim); tr.setDetector(None)
    tr.reserveHits(n * (lim.max_intersections + 1))
    tr.reset(); tr.setSurfaceSeed(s + 17)
    tr.traceRays(o, d); tr.sync()
    res[mode] = tr.hits()
res['oracle'] = oracle.trace_rays(sc, lim, o, d, nthreads=0, surface_seed=s + 17)['hits']
with oracle.strict():
  res['strict'] = oracle.trace_rays(sc, lim, o, d, nthreads=0, surface_seed=s + 17)['hits']
m48 = np.uint64(0xFFFFFFFFFFFF)
cnt = {k: np.bincount((v['tag'] & m48).astype(np.int64), minlength=n) for k, v in res.items()}
for a, b in (('1', '0'), ('1', 'oracle'), ('0', 'oracle'), ('oracle', 'strict')):
  bad = np.flatnonzero(cnt[a] != cnt[b])
  print(a, 'vs', b, 'rays with another number of hits:', len(bad), bad[:8].tolist())
bad = np.flatnonzero((cnt['1'] != cnt['oracle']) | (cnt['0'] != cnt['oracle']))
for b in bad[:3]:
  for k, v in res.items():
    h = v[(v['tag'] & m48).astype(np.int64) == b]
    print('ray', b, k, len(h))
    for row in h[:60]:
      print('    ', np.round(row['point'], 7).tolist(), np.round(row['direction'], 7).tolist(), int(row['tag'] >> np.uint64(48)) & 0x7fff, int(row['tag'] >> np.uint64(63)))
