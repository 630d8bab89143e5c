"""OpenCASCADE BRep text payloads (`<Object>.Shape.brp` inside an FCStd file).

Objects without a parametric recipe -- `Part::Feature` (STEP imports),
`PartDesign::Body` -- store only their boundary representation.  The
reference hands such shapes to OpenCASCADE (`shape.Faces`, `face.Surface`,
ray/face intersection through `Part.Line.intersect`, ray.py:353-430); without
FreeCAD this module reads the payload itself: the "CASCADE Topology V1" text
format BRepTools_ShapeSet writes (locations, 2-D / 3-D curves, surfaces,
TShapes), enough geometry evaluation to walk every face's trimming wires in
the parameter plane of its surface, and `scene/brep_mesh.py` turns each face
into facets for the tracer's triangle primitives.

Geometry kinds read (the ones the reference's own test files contain, plus
their obvious neighbours): curves -- line, circle, ellipse, B-spline
(rational or not), trimmed; surfaces -- plane, cylinder, cone, sphere, torus,
B-spline, rectangular-trimmed.  Anything else raises `BRepError` naming the
kind, so that nothing is approximated silently.
"""
from dataclasses import dataclass, field

import numpy as np
from scipy.interpolate import BSpline


class BRepError(ValueError):
  pass


# ------------------------------------------------------------------ locations
def _mat(rows=None):
  m = np.eye(4)
  if rows is not None:
    m[:3, :] = np.asarray(rows, dtype=np.float64).reshape(3, 4)
  return m


# ------------------------------------------------------------------ geometry
def _bspline_knots(knots, mults, degree=0, periodic=False):
  """flat knot vector; -> (flat knots, wrap): a periodic spline (OCC stores one period: n poles,
  first multiplicity = last) is unrolled into an ordinary one by continuing the knots `wrap` =
  degree + 1 - first multiplicity entries past both ends and repeating the first `wrap` poles
  (BSplCLib::KnotSequence / PoleIndex: the first span is controlled by poles 0..degree)"""
  knots, mults = np.asarray(knots, dtype=np.float64), np.asarray(mults, dtype=np.int64)
  flat = np.repeat(knots, mults)
  if not periodic:
    return flat, 0
  if mults[0] != mults[-1]:
    raise BRepError('periodic B-spline with different end multiplicities')
  wrap = max(0, degree + 1 - int(mults[0]))
  period = knots[-1] - knots[0]
  pre = flat[:len(flat) - mults[-1]][len(flat) - mults[-1] - wrap:] - period if wrap else flat[:0]
  post = flat[mults[0]:][:wrap] + period
  return np.concatenate([pre, flat, post]), wrap


def _wrap_poles(c, n_periodic, wrap, axis=0):
  idx = np.arange(n_periodic + wrap) % n_periodic
  return np.take(c, idx, axis=axis)


class Geometry:
  """curves: eval(t) -> (n, dim); surfaces: eval(u, v) -> (n, 3)"""
  kind = ''


class Line(Geometry):
  kind = 'line'

  def __init__(self, p, d):
    self.p, self.d = np.asarray(p, float), np.asarray(d, float)

  def eval(self, t):
    return self.p + np.asarray(t, float)[:, None] * self.d


class Conic(Geometry):
  """circle / ellipse: c + r1 cos(t) dx + r2 sin(t) dy"""

  def __init__(self, c, dx, dy, r1, r2, kind):
    self.c, self.dx, self.dy = np.asarray(c, float), np.asarray(dx, float), np.asarray(dy, float)
    self.r1, self.r2, self.kind = float(r1), float(r2), kind

  def eval(self, t):
    t = np.asarray(t, float)[:, None]
    return self.c + self.r1 * np.cos(t) * self.dx + self.r2 * np.sin(t) * self.dy


class BSplineCurve(Geometry):
  kind = 'bspline'

  def __init__(self, degree, poles, weights, knots, mults, periodic):
    t, wrap = _bspline_knots(knots, mults, degree, periodic)
    poles = np.asarray(poles, float)
    self.dim = poles.shape[1]
    self.rational = weights is not None
    c = np.hstack([poles * weights[:, None], weights[:, None]]) if self.rational else poles
    if periodic:
      c = _wrap_poles(c, len(c), wrap)
    if len(t) != len(c) + degree + 1:
      raise BRepError('inconsistent B-spline curve (knots / poles / degree)')
    self.t0, self.t1 = float(knots[0]), float(knots[-1])
    self.n_spans = len(knots) - 1
    self._s = BSpline(t, c, degree, extrapolate=True)

  def eval(self, t):
    v = self._s(np.clip(np.asarray(t, float), self.t0, self.t1))
    return v[:, :self.dim] / v[:, self.dim:] if self.rational else v


class TrimmedCurve(Geometry):
  def __init__(self, basis):
    self.basis, self.kind = basis, basis.kind

  def eval(self, t):
    return self.basis.eval(t)

  def __getattr__(self, key):
    return getattr(self.__dict__['basis'], key)


class Plane(Geometry):
  kind = 'plane'

  def __init__(self, p, n, dx, dy):
    self.p, self.n, self.dx, self.dy = (np.asarray(a, float) for a in (p, n, dx, dy))

  def eval(self, u, v):
    return self.p + np.asarray(u, float)[:, None] * self.dx + np.asarray(v, float)[:, None] * self.dy

  def normal(self, u, v):
    return np.broadcast_to(np.cross(self.dx, self.dy), (len(u), 3)).copy()

  def invert(self, x):
    d = np.asarray(x, float) - self.p
    return d @ self.dx, d @ self.dy

  def d1(self, u, v):
    """(dS/du, dS/dv), each (n, 3)"""
    n = len(np.atleast_1d(u))
    return np.tile(self.dx, (n, 1)), np.tile(self.dy, (n, 1))


class Revolved(Geometry):
  """cylinder / cone / sphere / torus in OCC's parametrisation (u = angle about n)"""

  def __init__(self, kind, p, n, dx, dy, r, extra=0.0):
    self.kind = kind
    self.p, self.n, self.dx, self.dy = (np.asarray(a, float) for a in (p, n, dx, dy))
    self.r, self.extra = float(r), float(extra)     # extra: cone semi-angle / torus minor radius

  def _frame(self, u):
    u = np.asarray(u, float)[:, None]
    return np.cos(u) * self.dx + np.sin(u) * self.dy

  def eval(self, u, v):
    e, v = self._frame(u), np.asarray(v, float)[:, None]
    if self.kind == 'cylinder':
      return self.p + self.r * e + v * self.n
    if self.kind == 'cone':
      return self.p + (self.r + v * np.sin(self.extra)) * e + v * np.cos(self.extra) * self.n
    if self.kind == 'sphere':
      return self.p + self.r * np.cos(v) * e + self.r * np.sin(v) * self.n
    return self.p + (self.r + self.extra * np.cos(v)) * e + self.extra * np.sin(v) * self.n     # torus

  def normal(self, u, v):
    """dS/du x dS/dv, normalised (analytic: well defined at the poles of a sphere too)"""
    e, v = self._frame(u), np.asarray(v, float)[:, None]
    hand = np.sign(np.dot(np.cross(self.dx, self.dy), self.n)) or 1.0
    if self.kind == 'cylinder':
      nrm = e
    elif self.kind == 'cone':
      nrm = np.cos(self.extra) * e - np.sin(self.extra) * self.n
      nrm = nrm * np.sign(self.r + v * np.sin(self.extra) + 1e-300)
    elif self.kind == 'sphere':
      nrm = np.cos(v) * e + np.sin(v) * self.n
    else:
      nrm = (np.cos(v) * e + np.sin(v) * self.n) * np.sign(self.r + self.extra * np.cos(v) + 1e-300)
    return nrm * hand

  def d1(self, u, v):
    """(dS/du, dS/dv), each (n, 3): exact"""
    u = np.asarray(u, float)[:, None]
    v = np.asarray(v, float)[:, None]
    e = np.cos(u) * self.dx + np.sin(u) * self.dy          # radial direction
    t = -np.sin(u) * self.dx + np.cos(u) * self.dy         # its derivative
    if self.kind == 'cylinder':
      return self.r * t, np.tile(self.n, (len(u), 1))
    if self.kind == 'cone':
      return (self.r + v * np.sin(self.extra)) * t, np.sin(self.extra) * e + np.cos(self.extra) * self.n
    if self.kind == 'sphere':
      return self.r * np.cos(v) * t, -self.r * np.sin(v) * e + self.r * np.cos(v) * self.n
    return (self.r + self.extra * np.cos(v)) * t, -self.extra * np.sin(v) * e + self.extra * np.cos(v) * self.n

  def invert(self, x, hint):
    """parameters of points x on the surface; angles are taken into the period around `hint`
    ((n, 2), e.g. the stored p-curve), which also stands in where u is undefined (poles, apex)"""
    d = np.asarray(x, float) - self.p
    z, ex, ey = d @ self.n, d @ self.dx, d @ self.dy
    rho = np.hypot(ex, ey)
    u = np.arctan2(ey, ex)
    u = hint[:, 0] + (u - hint[:, 0] + np.pi) % (2 * np.pi) - np.pi
    u = np.where(rho > 1e-9 * max(abs(self.r), 1.0), u, hint[:, 0])
    if self.kind == 'cylinder':
      v = z
    elif self.kind == 'cone':
      v = z / np.cos(self.extra)
    elif self.kind == 'sphere':
      v = np.arctan2(z, rho)
    else:
      v = np.arctan2(z, rho - self.r)
      v = hint[:, 1] + (v - hint[:, 1] + np.pi) % (2 * np.pi) - np.pi
    return np.stack([u, v], axis=1)

  def steps(self, tol):
    """parameter steps (du, dv) that keep the chord error below tol"""
    def ang(r):
      r = max(abs(r), 1e-12)
      return 2.0 * np.arccos(max(-1.0, 1.0 - min(tol / r, 1.0)))
    if self.kind == 'cylinder':
      return ang(self.r), np.inf
    if self.kind == 'cone':
      return None, np.inf          # radius varies along v: the mesher uses the largest one
    if self.kind == 'sphere':
      return ang(self.r), ang(self.r)
    return ang(abs(self.r) + abs(self.extra)), ang(self.extra)


class Parabola(Geometry):
  """gp_Parab / gp_Parab2d: vertex c, axis of symmetry dx, focal length f: c + t^2 / (4 f) dx + t dy"""
  kind = 'parabola'

  def __init__(self, c, dx, dy, focal):
    self.c, self.dx, self.dy, self.f = np.asarray(c, float), np.asarray(dx, float), np.asarray(dy, float), float(focal)

  def eval(self, t):
    t = np.asarray(t, float)[:, None]
    return self.c + t * t / (4.0 * self.f) * self.dx + t * self.dy


class Paraboloid(Geometry):
  """Geom_SurfaceOfRevolution of a parabola about its own axis of symmetry: S(u, v) = the point C(v) of the
  parabola turned by u about the axis (OpenCASCADE's parametrisation of a surface of revolution):
  p + v^2 / (4 f) n + v (cos u dx + sin u dy), p = vertex, n = axis towards the open side, dx = the parabola's
  own transverse direction, dy = axis x dx (axis = the direction of revolution, +n or -n)"""
  kind = 'paraboloid'

  def __init__(self, p, n, dx, dy, focal):
    self.p, self.n, self.dx, self.dy = (np.asarray(a, float) for a in (p, n, dx, dy))
    self.f = float(focal)

  def _frame(self, u):
    u = np.asarray(u, float)[:, None]
    return np.cos(u) * self.dx + np.sin(u) * self.dy

  def eval(self, u, v):
    e, v = self._frame(u), np.asarray(v, float)[:, None]
    return self.p + v * v / (4.0 * self.f) * self.n + v * e

  def d1(self, u, v):
    u = np.asarray(u, float)[:, None]
    v = np.asarray(v, float)[:, None]
    e = np.cos(u) * self.dx + np.sin(u) * self.dy
    t = -np.sin(u) * self.dx + np.cos(u) * self.dy
    return v * t, v / (2.0 * self.f) * self.n + e

  def normal(self, u, v):
    """dS/du x dS/dv normalised; at the vertex (v = 0) the limit: -+ the axis"""
    e, v = self._frame(u), np.asarray(v, float)[:, None]
    hand = np.sign(np.dot(np.cross(self.dx, self.dy), self.n)) or 1.0
    # t x (v / 2f n + e) with t = de/du:  (t x n) v / 2f + t x e = hand (e v / 2f - n)
    nrm = hand * (e * v / (2.0 * self.f) - self.n) * np.where(v < 0, -1.0, 1.0)
    return nrm / np.linalg.norm(nrm, axis=1, keepdims=True)

  def invert(self, x, hint):
    d = np.asarray(x, float) - self.p
    ex, ey = d @ self.dx, d @ self.dy
    rho = np.hypot(ex, ey)
    neg = hint[:, 1] < 0
    u = np.arctan2(ey, ex) + np.where(neg, np.pi, 0.0)
    u = hint[:, 0] + (u - hint[:, 0] + np.pi) % (2 * np.pi) - np.pi
    u = np.where(rho > 1e-9 * max(abs(self.f), 1.0), u, hint[:, 0])
    return np.stack([u, np.where(neg, -rho, rho)], axis=1)

  def steps_range(self, tol, lo, hi):
    """(du, dv) that keep the chord error below tol on lo <= (u, v) <= hi"""
    # (half the tolerance for the parallels, half for the meridians: the sags of a cell's two directions add up)
    rmax = max(abs(lo[1]), abs(hi[1]), 1e-12)
    du = 2.0 * np.arccos(max(-1.0, 1.0 - min(0.5 * tol / rmax, 1.0)))
    return du, 4.0 * np.sqrt(0.5 * tol * abs(self.f))       # curvature of the meridian <= 1 / (2 f)


class BSplineSurface(Geometry):
  kind = 'bspline-surface'

  def __init__(self, udeg, vdeg, poles, weights, uknots, umults, vknots, vmults, uperiodic, vperiodic):
    self.tu, wu = _bspline_knots(uknots, umults, udeg, uperiodic)
    self.tv, wv = _bspline_knots(vknots, vmults, vdeg, vperiodic)
    self.ku, self.kv = udeg, vdeg
    poles = np.asarray(poles, float)                 # (nu, nv, 3)
    self.rational = weights is not None
    c = np.concatenate([poles * weights[..., None], weights[..., None]], axis=2) if self.rational else poles
    if uperiodic:
      c = _wrap_poles(c, c.shape[0], wu, axis=0)
    if vperiodic:
      c = _wrap_poles(c, c.shape[1], wv, axis=1)
    self.c = c
    nu, nv = c.shape[:2]
    if len(self.tu) != nu + udeg + 1 or len(self.tv) != nv + vdeg + 1:
      raise BRepError('inconsistent B-spline surface (knots / poles / degree)')
    self.u0, self.u1 = float(uknots[0]), float(uknots[-1])
    self.v0, self.v1 = float(vknots[0]), float(vknots[-1])
    self.uperiodic, self.vperiodic = bool(uperiodic), bool(vperiodic)
    self.n_uspans, self.n_vspans = len(uknots) - 1, len(vknots) - 1

  def eval(self, u, v):
    u = np.clip(np.asarray(u, float), self.u0, self.u1)
    v = np.clip(np.asarray(v, float), self.v0, self.v1)
    nu, nv, d = self.c.shape
    bu = BSpline.design_matrix(u, self.tu, self.ku)           # (n, nu) sparse
    bv = BSpline.design_matrix(v, self.tv, self.kv).toarray() # (n, nv)
    x = (bu @ self.c.reshape(nu, nv * d)).reshape(len(u), nv, d)
    out = np.einsum('nj,njd->nd', bv, x)
    return out[:, :3] / out[:, 3:] if self.rational else out

  def normal(self, u, v):
    u, v = np.asarray(u, float), np.asarray(v, float)

    def cross(u, v):
      hu, hv = 1e-6 * (self.u1 - self.u0), 1e-6 * (self.v1 - self.v0)
      du = self.eval(np.minimum(u + hu, self.u1), v) - self.eval(np.maximum(u - hu, self.u0), v)
      dv = self.eval(u, np.minimum(v + hv, self.v1)) - self.eval(u, np.maximum(v - hv, self.v0))
      return np.cross(du, dv)
    n = cross(u, v)
    length = np.linalg.norm(n, axis=1)
    # where the parametrisation degenerates (the apex of a surface of revolution: a whole
    # boundary line is one point) the normal is the limit from just inside the patch
    bad = length < 1e-6 * max(length.max(), 1e-300)
    if bad.any():
      um, vm = 0.5 * (self.u0 + self.u1), 0.5 * (self.v0 + self.v1)
      n[bad] = cross(u[bad] + 1e-4 * (um - u[bad]), v[bad] + 1e-4 * (vm - v[bad]))
      length = np.linalg.norm(n, axis=1)
    return n / np.maximum(length, 1e-300)[:, None]


  def d1(self, u, v):
    """(dS/du, dS/dv): five-point differences (truncation and rounding both ~1e-13 relative)"""
    u, v = np.asarray(u, float), np.asarray(v, float)
    out = []
    for axis, (a, b) in enumerate(((self.u0, self.u1), (self.v0, self.v1))):
      h = 1e-3 * (b - a)
      x = u if axis == 0 else v
      c = np.clip(x, a + 2 * h, b - 2 * h)         # the stencil stays inside the patch
      ev = (lambda t: self.eval(t, v)) if axis == 0 else (lambda t: self.eval(u, t))
      d = (-ev(c + 2 * h) + 8 * ev(c + h) - 8 * ev(c - h) + ev(c - 2 * h)) / (12 * h)
      out.append(d)
    return out[0], out[1]


class TrimmedSurface(Geometry):
  def __init__(self, basis):
    self.basis, self.kind = basis, basis.kind

  def __getattr__(self, key):
    return getattr(self.__dict__['basis'], key)


# ------------------------------------------------------------------ topology
@dataclass
class TShape:
  kind: str                                   # Ve Ed Wi Fa Sh So CS Co
  subs: list = field(default_factory=list)    # [(orientation char, tshape index, location index)]
  # vertex
  point: np.ndarray = None
  tol: float = 0.0
  # edge: curve representations
  curve3d: tuple = None                       # (curve index, location index, first, last)
  pcurves: list = field(default_factory=list) # [(pcurve index, pcurve2 index or None, surface index, location index, first, last)]
  degenerated: bool = False
  # face
  surface: int = 0
  surface_loc: int = 0


class Payload:
  """a parsed BRep text: tables + the root shape reference"""

  def __init__(self, text):
    self._tok = text.split()
    self._i = 0
    head = text[:64]
    if 'CASCADE Topology V' not in head:
      raise BRepError('not an OpenCASCADE BRep text payload')
    self.version = int(head.split('Topology V')[1][0])
    if self.version > 3:
      raise BRepError(f'BRep format version {self.version} is not read (V1 - V3 are)')
    # (V2 adds the UV end points of p-curves, V3 normals to stored triangulations -- which are
    #  not read in any version)
    self._seek('Locations')
    self.locations = self._read_locations()
    self._seek('Curve2ds')
    self.curves2d = [self._read_curve(2) for _ in range(self._int())]
    self._seek('Curves')
    self.curves = [self._read_curve(3) for _ in range(self._int())]
    self._seek('Polygon3D')
    if self._int():
      raise BRepError('payloads with stored 3-D polygons are not read')
    self._seek('PolygonOnTriangulations')
    if self._int():
      raise BRepError('payloads with stored triangulations are not read')
    self._seek('Surfaces')
    self.surfaces = [self._read_surface() for _ in range(self._int())]
    self._seek('Triangulations')
    if self._int():
      raise BRepError('payloads with stored triangulations are not read')
    self._seek('TShapes')
    n = self._int()
    shapes = [self._read_tshape() for _ in range(n)]
    # the file lists TShape n first and TShape 1 last (BRepTools_ShapeSet::Write)
    self.tshapes = {n - k: s for k, s in enumerate(shapes)}
    self.root = self._read_sub()
    if self.root is None:
      raise BRepError('payload without a root shape')

  # -- token helpers ---------------------------------------------------------
  def _next(self):
    t = self._tok[self._i]
    self._i += 1
    return t

  def _int(self):
    return int(self._next())

  def _float(self):
    return float(self._next())

  def _floats(self, n):
    out = np.array(self._tok[self._i:self._i + n], dtype=np.float64)
    self._i += n
    return out

  def _seek(self, word):
    while self._tok[self._i] != word:
      self._i += 1
      if self._i >= len(self._tok):
        raise BRepError(f'section {word} not found')
    self._i += 1

  # -- sections --------------------------------------------------------------
  def _read_locations(self):
    n = self._int()
    raw = []
    for _ in range(n):
      t = self._int()
      if t == 1:
        raw.append(('m', _mat(self._floats(12))))
      elif t == 2:
        parts = []
        while True:
          idx = self._int()
          if idx == 0:
            break
          parts.append((idx, self._int()))
        raw.append(('c', parts))
      else:
        raise BRepError(f'location kind {t}')
    out = {0: np.eye(4)}
    for k, (kind, val) in enumerate(raw, start=1):
      if kind == 'm':
        out[k] = val
      else:
        m = np.eye(4)
        for idx, power in val:
          base = out[idx]
          m = m @ np.linalg.matrix_power(np.linalg.inv(base) if power < 0 else base, abs(power))
        out[k] = m
    return out

  def _read_bspline_curve(self, dim):
    rational, periodic, degree, npoles, nknots = (self._int() for _ in range(5))
    poles, weights = np.empty((npoles, dim)), (np.empty(npoles) if rational else None)
    for i in range(npoles):
      poles[i] = self._floats(dim)
      if rational:
        weights[i] = self._float()
    knots, mults = np.empty(nknots), np.empty(nknots, dtype=np.int64)
    for i in range(nknots):
      knots[i], mults[i] = self._float(), self._int()
    return BSplineCurve(degree, poles, weights, knots, mults, periodic)

  def _read_curve(self, dim):
    t = self._int()
    if t == 1:
      return Line(self._floats(dim), self._floats(dim))
    if t in (2, 3):
      c = self._floats(dim)
      if dim == 3:
        self._floats(3)                      # axis
      dx, dy = self._floats(dim), self._floats(dim)
      r1 = self._float()
      r2 = self._float() if t == 3 else r1
      return Conic(c, dx, dy, r1, r2, 'circle' if t == 2 else 'ellipse')
    if t == 4:                               # parabola: vertex, [normal of its plane,] axis of symmetry, transverse, focal
      c = self._floats(dim)
      if dim == 3:
        self._floats(3)
      dx, dy = self._floats(dim), self._floats(dim)
      return Parabola(c, dx, dy, self._float())
    if t == 7:
      return self._read_bspline_curve(dim)
    if t == 8:
      self._floats(2)
      return TrimmedCurve(self._read_curve(dim))
    raise BRepError(f'{dim}-D curve kind {t} (hyperbola / Bezier / offset) is not read')

  def _read_surface(self):
    t = self._int()
    if t == 1:
      return Plane(self._floats(3), self._floats(3), self._floats(3), self._floats(3))
    if t in (2, 3, 4, 5):
      p, n, dx, dy = (self._floats(3) for _ in range(4))
      r = self._float()
      extra = self._float() if t in (3, 5) else 0.0
      return Revolved({2: 'cylinder', 3: 'cone', 4: 'sphere', 5: 'torus'}[t], p, n, dx, dy, r, extra)
    if t == 9:
      urat, vrat, uper, vper, udeg, vdeg, nu, nv, nuk, nvk = (self._int() for _ in range(10))
      rational = bool(urat or vrat)
      poles, weights = np.empty((nu, nv, 3)), (np.empty((nu, nv)) if rational else None)
      for i in range(nu):
        for j in range(nv):
          poles[i, j] = self._floats(3)
          if rational:
            weights[i, j] = self._float()
      uk, um = np.empty(nuk), np.empty(nuk, dtype=np.int64)
      for i in range(nuk):
        uk[i], um[i] = self._float(), self._int()
      vk, vm = np.empty(nvk), np.empty(nvk, dtype=np.int64)
      for i in range(nvk):
        vk[i], vm[i] = self._float(), self._int()
      return BSplineSurface(udeg, vdeg, poles, weights, uk, um, vk, vm, uper, vper)
    if t == 7:
      # surface of revolution: point and direction of the axis, then the basis curve.  Read when it is a
      # paraboloid -- a parabola turned about its own axis of symmetry (parabolic mirrors, README "slotted
      # parabolic mirrors") --, which the tracer knows as an analytic primitive
      a, d = self._floats(3), self._floats(3)
      c = self._read_curve(3)
      basis = c.basis if isinstance(c, TrimmedCurve) else c
      if basis.kind != 'parabola':
        raise BRepError(f'surface of revolution of a {basis.kind} is not read (only of a parabola about its axis)')
      d = d / np.linalg.norm(d)
      n = basis.dx / np.linalg.norm(basis.dx)
      off = basis.c - a
      if abs(abs(n @ d) - 1.0) > 1e-9 or np.linalg.norm(off - (off @ d) * d) > 1e-9 * max(1.0, abs(basis.f)):
        raise BRepError('surface of revolution of a parabola about another line than its axis is not read')
      dx = basis.dy / np.linalg.norm(basis.dy)
      return Paraboloid(basis.c, n, dx, np.cross(d, dx), basis.f)
    if t == 10:
      self._floats(4)
      return TrimmedSurface(self._read_surface())
    raise BRepError(f'surface kind {t} (extrusion / Bezier / offset) is not read')

  def _read_sub(self):
    t = self._next()
    if t == '*':
      return None
    if t[0] not in '+-ie':
      raise BRepError(f'unexpected token {t!r} in a sub-shape list')
    return (t[0], int(t[1:]), self._int())

  def _read_tshape(self):
    kind = self._next()
    s = TShape(kind)
    if kind == 'Ve':
      s.tol = self._float()
      s.point = self._floats(3)
      while True:                             # point representations "param kind ...", closed by "0 0"
        self._float()
        t = self._int()
        if t == 0:
          break
        if t == 1:
          self._int(); self._int()
        elif t == 2:
          self._int(); self._int(); self._int()
        elif t == 3:
          self._float(); self._int(); self._int()
        else:
          raise BRepError(f'vertex representation kind {t}')
    elif kind == 'Ed':
      s.tol = self._float()
      self._int(); self._int()
      s.degenerated = bool(self._int())
      while True:
        t = self._int()
        if t == 0:
          break
        if t == 1:
          s.curve3d = (self._int(), self._int(), self._float(), self._float())
        elif t == 2:
          pc, surf, loc = self._int(), self._int(), self._int()
          first, last = self._float(), self._float()
          if self.version >= 2:
            self._floats(4)                   # UV of the two ends (written since format version 2)
          s.pcurves.append((pc, None, surf, loc, first, last))
        elif t == 3:
          pc = self._int()
          t2 = self._next()                   # "<pcurve 2><continuity>" are written without a blank
          digits = ''.join(ch for ch in t2 if ch.isdigit())
          if len(digits) == len(t2):
            self._next()
          pc2 = int(digits)
          surf, loc = self._int(), self._int()
          first, last = self._float(), self._float()
          if self.version >= 2:
            self._floats(4)                   # UV points of the second p-curve (UVPoints2)
          s.pcurves.append((pc, pc2, surf, loc, first, last))
        elif t == 4:
          self._next(); self._int(); self._int(); self._int(); self._int()
        else:
          raise BRepError(f'edge representation kind {t} (polygons) is not read')
    elif kind == 'Fa':
      self._int()
      s.tol = self._float()
      s.surface, s.surface_loc = self._int(), self._int()
      if self._tok[self._i] == '2':           # triangulation reference
        self._int(); self._int()
    elif kind not in ('Wi', 'Sh', 'So', 'CS', 'Co'):
      raise BRepError(f'unknown TShape kind {kind!r}')
    flags = self._next()
    if len(flags) != 7 or set(flags) - {'0', '1'}:
      raise BRepError(f'malformed TShape flags {flags!r}')
    while True:
      sub = self._read_sub()
      if sub is None:
        break
      s.subs.append(sub)
    return s

  # -- traversal ---------------------------------------------------------------
  def faces(self):
    """-> [(face tshape index, 4x4 location of the face, reversed?)] in explorer
    order (depth first, sub-shapes in stored order; `Face1` is the first one)"""
    out, seen = [], set()

    def walk(ref, loc, rev):
      o, idx, l = ref
      ts = self.tshapes[idx]
      here = loc @ self.locations[l]
      r = rev ^ (o == '-')
      if ts.kind == 'Fa':
        key = (idx, tuple(np.round(here, 12).ravel()), r)
        if key not in seen:
          seen.add(key)
          out.append((idx, here, r))
        return
      for sub in ts.subs:
        walk(sub, here, r)

    walk(self.root, np.eye(4), False)
    return out


def load(text):
  if isinstance(text, bytes):
    text = text.decode('latin1')
  return Payload(text)
