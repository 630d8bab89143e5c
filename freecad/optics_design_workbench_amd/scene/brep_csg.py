"""BRep solids that are intersections of quadric half-spaces, as exact CSG.

Catalogue lenses arrive as STEP files: two spheres and a cylinder, sometimes
planes, cones or tori.  The reference intersects rays with exactly these
surfaces (OpenCASCADE, ray.py:353-430).  A tessellation (scene/brep_mesh.py) is
the general answer here, but when a solid is bounded by planes, spheres,
cylinders, cones and tori only, and every face is the whole part of its
surface that lies on the material side of all the other surfaces, the solid is

    intersection over its surfaces of  { inside S } or { outside S }

and the tracer's analytic primitives describe it without any approximation:
the Common of the "inside" primitives, Cut by the "outside" ones (Part::Common /
Part::Cut semantics of scene/geometry.py).  Planes become one face of a box
that extends far to the material side, unbounded cylinders and cones finite
ones that reach beyond the solid; the faces such auxiliary shapes add lie
outside the other primitives and are masked out.

`recognise` returns the CSG tree or None -- None whenever the hypothesis does
not hold on a sampled check (non-convex arrangements such as an L-shaped
prism, faces trimmed by edges that are not surface intersections, B-spline
surfaces): those shapes keep their facets.
"""
import numpy as np

from . import brep_mesh
from .placement import Placement

BOX, SPHERE, CYLINDER, CONE, TORUS = range(5)
PARABOLOID = 6


def _frame(origin, z, x=None):
  """Placement with local z along `z` (and x near `x`) at `origin`"""
  z = np.asarray(z, float) / np.linalg.norm(z)
  if x is None or abs(np.dot(x, z)) > 0.999 * np.linalg.norm(x):
    x = np.array([1.0, 0, 0]) if abs(z[0]) < 0.9 else np.array([0, 1.0, 0])
  x = np.asarray(x, float) - np.dot(x, z) * z
  x /= np.linalg.norm(x)
  m = np.eye(4)
  m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, np.cross(z, x), z, origin
  return Placement(matrix=m)


class _Surface:
  """one bounding surface in the coordinates of the shape: implicit function f (negative on the
  side the analytic primitive calls inside) and its gradient direction"""

  def __init__(self, kind, p, axis, xdir, r=0.0, extra=0.0):
    self.kind, self.p, self.axis, self.xdir, self.r, self.extra = kind, p, axis, xdir, r, extra

  def key(self):
    return (self.kind,) + tuple(np.round(np.concatenate([self.p, self.axis, [self.r, self.extra]]), 7))

  def f(self, x):
    d = x - self.p
    z = d @ self.axis
    if self.kind == 'plane':
      return z
    rho = np.linalg.norm(d - z[:, None] * self.axis, axis=1)
    if self.kind == 'sphere':
      return np.linalg.norm(d, axis=1) - self.r
    if self.kind == 'cylinder':
      return rho - self.r
    if self.kind == 'cone':
      return (rho - (self.r + z * np.tan(self.extra))) * np.cos(self.extra)
    if self.kind == 'paraboloid':                          # rho^2 = 4 f z, r = f: distance to first order
      return (rho * rho - 4.0 * self.r * z) / (2.0 * np.sqrt(rho * rho + 4.0 * self.r * self.r))
    return np.hypot(rho - self.r, z) - self.extra          # torus

  def grad(self, x, h=1e-6):
    g = np.stack([(self.f(x + h * e) - self.f(x - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
    return g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-300)


def _surface_of(s, loc):
  """brep surface + its location -> _Surface in shape coordinates (None: not a quadric / torus)"""
  if s.kind not in ('plane', 'sphere', 'cylinder', 'cone', 'torus', 'paraboloid'):
    return None
  R = loc[:3, :3]
  p = R @ s.p + loc[:3, 3]
  if s.kind == 'paraboloid':
    axis = R @ s.n
    return _Surface('paraboloid', p, axis / np.linalg.norm(axis), R @ s.dx, abs(s.f))
  if s.kind == 'plane':
    n = R @ np.cross(s.dx, s.dy)
    return _Surface('plane', p, n / np.linalg.norm(n), R @ s.dx)
  axis = R @ s.n
  out = _Surface(s.kind, p, axis / np.linalg.norm(axis), R @ s.dx, abs(s.r), s.extra)
  if s.kind == 'cone' and s.r < 0:
    return None
  return out


def recognise(payload, mesh, tol=1e-6, samples=48):
  """-> (tree of geometry.Node, number of surfaces) or None.  `mesh` = brep_mesh.tessellate(payload,
  keep_root_location=False) (its facets provide the sample points and the extent)."""
  from .geometry import Node
  P = payload
  base = np.linalg.inv(P.locations[P.root[2]])
  solids = [s for s in P.tshapes.values() if s.kind == 'So']
  if len(solids) != 1:
    return None
  surfaces, faces_of = [], {}
  for k, (fidx, loc, rev) in enumerate(P.faces()):
    f = P.tshapes[fidx]
    s = _surface_of(P.surfaces[f.surface - 1], base @ loc @ P.locations[f.surface_loc])
    if s is None:
      return None
    key = s.key()
    if key not in faces_of:
      faces_of[key] = (len(surfaces), [])
      surfaces.append(s)
    faces_of[key][1].append(k)
  if len(surfaces) > 24:
    return None
  lo, hi = mesh.vertices.min(axis=0), mesh.vertices.max(axis=0)
  size = float(np.linalg.norm(hi - lo))
  centre = 0.5 * (lo + hi)
  # side of every surface the material is on: outward face normal against the gradient of f
  sign = np.zeros(len(surfaces))
  face_pts = {}
  for key, (si, ks) in faces_of.items():
    votes = []
    for k in ks:
      fm = mesh.faces[k]
      tri = mesh.triangles[fm.first:fm.first + fm.count]
      vid = np.unique(tri)
      vid = vid[:: max(1, len(vid) // 400)]
      x, n = mesh.vertices[vid], mesh.normals[vid]
      face_pts.setdefault(si, []).append(x)
      if np.abs(surfaces[si].f(x)).max() > 1e-6 * max(1.0, size):
        return None
      votes.append(np.einsum('ij,ij->i', n, surfaces[si].grad(x)))
    votes = np.concatenate(votes)
    if np.abs(votes).min() < 0.5 or (votes > 0).any() == (votes < 0).any():
      return None                                    # one surface bounding the material from both sides
    sign[si] = 1.0 if votes[0] > 0 else -1.0         # +1: material where f < 0 (inside the primitive)
  # necessary: every face lies on the material side of all the other surfaces
  for si, pts in face_pts.items():
    x = np.concatenate(pts)
    for sj, s in enumerate(surfaces):
      if sj != si and (sign[sj] * s.f(x)).max() > tol * max(1.0, size):
        return None
  # sufficient: inside the bounding box of the shape, points of a surface that satisfy all the
  # others belong to one of its faces (sampled over the surface's whole extent in the box: an
  # intersection of half-spaces can have further components elsewhere, e.g. beyond the far side
  # of a sphere -- the box, added to the Common below without faces of its own, shuts them out).
  # Wires as polygons of 2e-4 x size chord error; samples closer than 1e-3 x size to another
  # surface or to the box are not judged.
  m = brep_mesh._Mesher(P, 2e-4 * max(size, 1e-9), 64)
  margin = max(10 * tol * max(1.0, size), 1e-3 * size)
  pad = 0.05 * size + 10 * margin
  corners = np.array([[x, y, z] for x in (lo[0] - pad, hi[0] + pad) for y in (lo[1] - pad, hi[1] + pad)
                      for z in (lo[2] - pad, hi[2] + pad)])
  for key, (si, ks) in faces_of.items():
    faces = []
    for k in ks:
      fidx, loc, rev = P.faces()[k]
      f = P.tshapes[fidx]
      surf = P.surfaces[f.surface - 1]
      loc_surf = base @ loc @ P.locations[f.surface_loc]
      loops, _ = m._loops(f, base @ loc, surf, np.linalg.inv(loc_surf), 8)
      faces.append((surf, loc_surf, loops))
    surf, loc_surf, _ = faces[0]
    local = brep_mesh._xf(np.linalg.inv(loc_surf), corners)
    n_s = 2 * samples
    if surf.kind == 'plane':
      cu, cv = surf.invert(local)
      us, vs = (c.min() + (c.max() - c.min()) * (np.arange(n_s) + 0.5) / n_s for c in (cu, cv))
    else:
      us = 2 * np.pi * (np.arange(n_s) + 0.5) / n_s
      if surf.kind == 'sphere':
        vs = -np.pi / 2 + np.pi * (np.arange(n_s) + 0.5) / n_s
      elif surf.kind == 'torus':
        vs = us.copy()
      elif surf.kind == 'paraboloid':
        # v = distance from the axis: as far as the box reaches
        d = local - surf.p
        rmax = np.linalg.norm(d - (d @ surf.n)[:, None] * surf.n, axis=1).max()
        vs = rmax * (np.arange(n_s) + 0.5) / n_s
      else:
        z = (local - surf.p) @ surf.n
        z = z / np.cos(surf.extra) if surf.kind == 'cone' else z
        vs = z.min() + (z.max() - z.min()) * (np.arange(n_s) + 0.5) / n_s
    g = np.stack(np.meshgrid(us, vs, indexing='ij'), axis=-1).reshape(-1, 2)
    x = brep_mesh._xf(loc_surf, surf.eval(g[:, 0], g[:, 1]))
    ok = ((x > lo - pad + margin) & (x < hi + pad - margin)).all(axis=1)
    for sj, s in enumerate(surfaces):
      if sj != si:
        ok &= sign[sj] * s.f(x) < -margin
    rest = x[ok]
    for surf2, loc2, loops2 in faces:
      if not len(rest):
        break
      uv_all = np.concatenate(loops2)[:, :2]
      local2 = brep_mesh._xf(np.linalg.inv(loc2), rest)
      if surf2.kind == 'plane':
        uv2 = np.stack(surf2.invert(local2), axis=1)
      else:
        hint = np.broadcast_to(0.5 * (uv_all.min(axis=0) + uv_all.max(axis=0)), (len(rest), 2))
        uv2 = surf2.invert(local2, hint)
      rest = rest[~brep_mesh._inside(uv2, loops2)]
    if len(rest):
      return None
  # the tree: Common of the primitives the material is inside of, Cut by the others
  far = 4.0 * size + 1.0
  inside, outside = [], []
  for s, sg in zip(surfaces, sign):
    node = _primitive(s, sg, centre, size, far)
    if node is None:
      return None
    (inside if sg > 0 or s.kind == 'plane' else outside).append(node)
  if not inside:
    return None
  if len(surfaces) > 1:
    # the bounding box of the check above: a Common operand without faces of its own
    inside.append(Node('prim', placement=Placement(base=lo - pad), kind=BOX,
                       params=tuple(float(v) for v in (hi - lo + 2 * pad)) + (0.0,), facemask=0))
  tree = inside[0] if len(inside) == 1 else Node('common', children=inside)
  for tool in outside:
    tree = Node('cut', children=[tree, tool])
  return tree, len(surfaces)


def _primitive(s, sg, centre, size, far):
  from .geometry import Node
  if s.kind == 'sphere':
    return Node('prim', placement=Placement(base=s.p), kind=SPHERE, params=(s.r, 0.0, 0.0, 0.0))
  if s.kind == 'torus':
    return Node('prim', placement=_frame(s.p, s.axis, s.xdir), kind=TORUS, params=(s.r, s.extra, 0.0, 0.0))
  if s.kind == 'paraboloid':
    # x^2 + y^2 <= 4 f z up to a height beyond the solid (its cap is masked out: faces bit 0 only)
    h = max((centre - s.p) @ s.axis, 0.0) + far
    return Node('prim', placement=_frame(s.p, s.axis, s.xdir), kind=PARABOLOID, params=(s.r, h, 2.0 * np.sqrt(s.r * h), 0.0),
                facemask=1)
  if s.kind == 'plane':
    # one face of a box that extends `far` to the material side: the face z = far of a box whose
    # local z runs along the outward normal (f < 0 is behind the plane; sg < 0 turns it round)
    n = s.axis * (1.0 if sg > 0 else -1.0)
    foot = centre - ((centre - s.p) @ s.axis) * s.axis
    pl = _frame(foot - far * n, n, s.xdir)
    pl = pl * Placement(base=(-far, -far, 0.0))
    return Node('prim', placement=pl, kind=BOX, params=(2 * far, 2 * far, far, 0.0), facemask=1 << 5)
  z0 = (centre - s.p) @ s.axis
  if s.kind == 'cylinder':
    pl = _frame(s.p + (z0 - far) * s.axis, s.axis, s.xdir)
    return Node('prim', placement=pl, kind=CYLINDER, params=(s.r, 2 * far, 0.0, 0.0), facemask=1)
  # cone: the part of it around the solid where its radius is positive
  t = np.tan(s.extra)
  za, zb = z0 - far, z0 + far
  if abs(t) > 1e-12:
    apex = -s.r / t
    if t > 0:
      za = max(za, apex + 1e-6 * far)
    else:
      zb = min(zb, apex - 1e-6 * far)
  if zb <= za:
    return None
  pl = _frame(s.p + za * s.axis, s.axis, s.xdir)
  return Node('prim', placement=pl, kind=CONE, params=(s.r + za * t, s.r + zb * t, zb - za, 0.0), facemask=1)
