#!/usr/bin/env python3
"""Randomised check of the mesh kernel's normal cones (odw_capi.hip: WideBvh::cone_word; odw_mesh.hip: `inside`): convex
hulls of random point clouds -- blobs, needles, discs, cut balls; a few hundred to a few ten thousand facets of every shape
and size ratio -- as lenses of random index and as mirrors under a wide beam, distTol drawn from 1e-6 .. 1e-2.  The rows of
the mesh kernel with cones must be those without (ODW_MESH_CONES=0) and those of the binary-tree kernel (ODW_MESH_KERNEL=0),
bit for bit.
  python tests/fuzz_cones.py [n_scenes] [rays] [seed]
Prints one JSON line; exit code 1 if anything differs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.spatial import ConvexHull

from freecad.optics_design_workbench_amd.freecad_elements import make, point_source
from freecad.optics_design_workbench_amd.scene import Document, bake, geometry
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)


def hull(kind, n):
  p = rng.normal(size=(n, 3))
  if kind == 0:                                  # points on a sphere: every point a vertex
    p /= np.linalg.norm(p, axis=1)[:, None]
  elif kind == 1:                                # a needle / a disc: an ellipsoid of random axes
    p /= np.linalg.norm(p, axis=1)[:, None]
    p *= 10.0 ** rng.uniform(-1.2, 0.3, size=3)
  elif kind == 2:                                # a cut ball: flat faces with many coplanar facets
    p /= np.linalg.norm(p, axis=1)[:, None]
    p[:, 2] = np.clip(p[:, 2], -rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8))
  else:                                          # a blob: few of the points are vertices, facets of every size
    p *= rng.uniform(0.3, 1.0, size=(n, 1))
  h = ConvexHull(p)
  tri = h.simplices.copy()
  # counter-clockwise seen from outside
  c = p[h.vertices].mean(axis=0)
  nrm = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
  flip = np.einsum('ij,ij->i', nrm, p[tri[:, 0]] - c) < 0
  tri[flip] = tri[flip][:, ::-1]
  used = np.unique(tri)
  remap = -np.ones(len(p), dtype=np.int64)
  remap[used] = np.arange(len(used))
  return p[used] * rng.uniform(2.0, 6.0), remap[tri]


bad, done, strict = 0, 0, 0
for s in range(n_scenes):
  v, tri = hull(s % 4, int(10 ** rng.uniform(1.5, 4.2)))
  doc = Document()
  ang = rng.uniform(0, np.pi)
  mesh = make.makeMesh(doc, v, tri, base=(rng.uniform(-1, 1), rng.uniform(-1, 1), 25.0), quat=(np.sin(ang / 2), 0, 0, np.cos(ang / 2)))
  if s % 5 == 4:
    make.makeMirror(doc, [mesh], RecordHits=True)
  else:
    make.makeLens(doc, [mesh], RefractiveIndex=float(rng.uniform(1.2, 2.8)), RecordHits=True)
  make.makeAbsorber(doc, [make.makeBox(doc, 'A', 400, 400, 1, base=(-200, -200, 90))])
  make.makeSimulationSettings(doc, MaxIntersections=30.0, DistanceTolerance=repr(float(10 ** rng.uniform(-6, -2))))
  src = make.makePointSource(doc, PowerDensity='1', ThetaDomain='0, 0.3')
  sc, lim, bs = bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)
  tri_rows = sc.prim_type == geometry.TRIANGLE
  if not (sc.prim_flags[tri_rows] & 2).all():
    continue                                     # (a hull the bake does not call convex: rounding at a sliver; nothing to test)
  strict += bool((sc.prim_flags[tri_rows] & 8).all())
  rows = {}
  for mode, kernel, cones in (('cones', '1', '1'), ('plain', '1', '0'), ('binary', '0', '1')):
    os.environ['ODW_MESH_KERNEL'], os.environ['ODW_MESH_CONES'] = kernel, cones
    with Tracer(0) as tr:
      tr.setScene(sc); tr.setSource(bs); tr.setLimits(lim); tr.setDetector(None)
      tr.reserveHits(rays * 8)
      tr.reset()
      tr.trace(7 * s, rays, seed + s, histogram=False)
      tr.sync()
      rows[mode] = (tr.counters(), tr.hits())
  done += 1
  for other in ('plain', 'binary'):
    same = rows['cones'][0] == rows[other][0] and all(np.array_equal(rows['cones'][1][c], rows[other][1][c]) for c in ('tag', 'point', 'direction', 'power'))
    if not same:
      bad += 1
      print(json.dumps(dict(scene=s, against=other, facets=int(tri_rows.sum()), dist_tol=lim.dist_tol, counters=rows['cones'][0], other=rows[other][0])), flush=True)
print(json.dumps(dict(scenes=done, strictly_convex=strict, rays_each=rays, differing=bad)))
sys.exit(1 if bad else 0)
