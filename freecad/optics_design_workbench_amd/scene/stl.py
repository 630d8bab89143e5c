"""STL meshes (binary and ASCII) -> vertex / triangle arrays.

The route for geometry the FCStd-lite bake cannot express (BRep imports,
B-spline faces): export the shape from any CAD program as STL, or let a
FreeCAD-side binding pass `Shape.tessellate(tol)`.  STL
stores float32 facets without connectivity; `weld` merges coincident
vertices, `smoothNormals` derives vertex normals for curved faces.
"""
import struct

import numpy as np


def readSTL(path, weldTol=0.0):
  """-> (vertices (n,3) float64, triangles (m,3) int64), coincident vertices
  welded (exactly equal, or equal after rounding to `weldTol`)"""
  data = open(path, 'rb').read()
  tri = None
  if len(data) >= 84:
    (count,) = struct.unpack_from('<I', data, 80)
    if 84 + 50 * count == len(data):          # binary: 80-byte header, count, 50-byte records
      rec = np.frombuffer(data, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]), count=count,
                          offset=84)
      tri = rec['v'].astype(np.float64)
  if tri is None:
    vs = []
    for line in data.decode('ascii', errors='replace').splitlines():
      parts = line.split()
      if len(parts) == 4 and parts[0] == 'vertex':
        vs.append([float(x) for x in parts[1:]])
    if not vs or len(vs) % 3:
      raise ValueError(f'{path}: not an STL file')
    tri = np.array(vs, dtype=np.float64).reshape(-1, 3, 3)
  return weld(tri.reshape(-1, 3), np.arange(len(tri) * 3).reshape(-1, 3), weldTol)


def weld(vertices, triangles, tol=0.0):
  """merge coincident vertices (exactly equal, or equal after rounding to `tol`)"""
  v = np.asarray(vertices, dtype=np.float64)
  key = v if tol <= 0 else np.round(v / tol)
  _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
  tri = inverse.reshape(-1)[np.asarray(triangles, dtype=np.int64)]
  keep = (tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])
  return v[first], tri[keep]


def smoothNormals(vertices, triangles, creaseAngleDeg=30.0):
  """-> (vertices', triangles', vertex normals'): angle-weighted vertex normals;
  a vertex shared by facets whose normals differ by more than the crease angle
  is split, so edges (cylinder rims, box edges) stay sharp"""
  v = np.asarray(vertices, dtype=np.float64)
  tri = np.asarray(triangles, dtype=np.int64)
  p = v[tri]
  fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
  fn /= np.linalg.norm(fn, axis=1)[:, None]
  # corner angles as weights
  ang = np.empty((len(tri), 3))
  for k in range(3):
    a, b = p[:, (k + 1) % 3] - p[:, k], p[:, (k + 2) % 3] - p[:, k]
    c = np.einsum('ij,ij->i', a, b) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    ang[:, k] = np.arccos(np.clip(c, -1, 1))
  cos_crease = np.cos(np.radians(creaseAngleDeg))
  order = np.argsort(tri.reshape(-1), kind='stable')
  corner_vertex = tri.reshape(-1)[order]
  starts = np.concatenate([[0], np.nonzero(np.diff(corner_vertex))[0] + 1, [len(order)]])
  new_v, new_n = [], []
  new_tri = tri.copy()
  for s, e in zip(starts[:-1], starts[1:]):
    corners = order[s:e]                          # flat corner ids (facet*3 + k) around one vertex
    facets = corners // 3
    todo = list(range(len(corners)))
    while todo:                                    # greedy clusters of facets within the crease angle
      seed = todo[0]
      members = [j for j in todo if fn[facets[j]] @ fn[facets[seed]] >= cos_crease]
      todo = [j for j in todo if j not in members]
      w = ang.reshape(-1)[corners[members]]
      nn = (fn[facets[members]] * w[:, None]).sum(axis=0)
      nn /= np.linalg.norm(nn)
      new_tri.reshape(-1)[corners[members]] = len(new_v)
      new_v.append(v[corner_vertex[s]])
      new_n.append(nn)
  return np.array(new_v), new_tri, np.array(new_n)


def writeSTL(path, vertices, triangles):
  """binary STL (tests and round trips)"""
  v = np.asarray(vertices, dtype=np.float64)[np.asarray(triangles, dtype=np.int64)]
  fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
  fn /= np.maximum(np.linalg.norm(fn, axis=1), 1e-300)[:, None]
  rec = np.zeros(len(v), dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]))
  rec['n'], rec['v'] = fn, v
  with open(path, 'wb') as f:
    f.write(b'binary STL'.ljust(80, b' '))
    f.write(struct.pack('<I', len(v)))
    f.write(rec.tobytes())
