"""Facets of a BRep payload: what `Shape.tessellate(tol)` returns in FreeCAD.

Every face is meshed in the parameter plane of its surface:
  1. the trimming wires become closed polylines in (u, v): the edges' stored
     p-curves sampled at parameters shared by both faces of an edge (so the two
     faces meet in the same 3-D points), planar faces without stored p-curves
     by projecting the 3-D curves onto the plane;
  2. curved surfaces get interior points on a (u, v) grid whose steps keep the
     chord error below the deflection;
  3. a Delaunay triangulation of all points, made conforming by splitting
     boundary segments that are not edges of it, keeps the triangles whose
     centroid lies inside the wires (even-odd rule: holes come for free);
  4. vertices and unit normals are evaluated on the exact surface (dS/du x
     dS/dv, reversed for reversed faces: out of the solid), so the tracer's
     normal interpolation sees the true surface normals at the corners.
"""
from dataclasses import dataclass

import numpy as np
from scipy.spatial import Delaunay, cKDTree

from . import brep
from .brep import BRepError


@dataclass
class FaceMesh:
  index: int                 # 1-based: `Face<index>` in FreeCAD's naming
  kind: str                  # plane / cylinder / cone / sphere / torus / bspline-surface
  first: int                 # slice of the triangle list
  count: int
  area: float


@dataclass
class ShapeMesh:
  vertices: np.ndarray       # (n, 3)
  normals: np.ndarray        # (n, 3) unit, out of the solid
  triangles: np.ndarray      # (m, 3) counter-clockwise seen from outside
  faces: list                # [FaceMesh]

  def volume(self):
    a, b, c = (self.vertices[self.triangles[:, k]] for k in range(3))
    return float(np.einsum('ij,ij->i', a, np.cross(b, c)).sum() / 6.0)

  def area(self):
    a, b, c = (self.vertices[self.triangles[:, k]] for k in range(3))
    return float(0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum())

  def open_edges(self, tol=1e-7):
    """edges (by welded end points) that are not shared by exactly two facets"""
    key = np.round(self.vertices / tol).astype(np.int64)
    _, ids = np.unique(key, axis=0, return_inverse=True)
    t = ids.ravel()[self.triangles] if ids.ndim > 1 else ids[self.triangles]
    t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    return int((counts != 2).sum())


def _xf(m, p):
  return p @ m[:3, :3].T + m[:3, 3]


def _circle_segments(radius, span, tol):
  r = max(abs(radius), 1e-12)
  step = 2.0 * np.arccos(max(-1.0, 1.0 - min(tol / r, 1.0)))
  return max(1, int(np.ceil(abs(span) / max(step, 1e-6))))


class _Mesher:

  def __init__(self, payload, deflection, max_grid):
    self.P = payload
    self.tol = float(deflection)
    self.max_grid = int(max_grid)
    self._edge_n = {}
    self._edge_s = {}        # edge -> normalised sample parameters in [0, 1], shared by both faces
    self.requests = []       # (edge, s): boundary chords a face had to split (refined for all faces next pass)

  # -- edges -----------------------------------------------------------------
  def edge_segments(self, eidx):
    """number of chords of an edge: a property of the edge, so that the faces on both sides agree"""
    if eidx in self._edge_n:
      return self._edge_n[eidx]
    e = self.P.tshapes[eidx]
    n = 0
    if e.curve3d is not None and not e.degenerated:
      c = self.P.curves[e.curve3d[0] - 1]
      first, last = e.curve3d[2], e.curve3d[3]
      if c.kind == 'line':
        n = 1
      elif c.kind in ('circle', 'ellipse'):
        n = _circle_segments(max(c.r1, c.r2), last - first, self.tol)
      else:
        n = 8
        while n < 2048:
          t = np.linspace(first, last, n + 1)
          mid = c.eval(0.5 * (t[1:] + t[:-1]))
          ends = c.eval(t)
          if np.linalg.norm(mid - 0.5 * (ends[1:] + ends[:-1]), axis=1).max() < self.tol:
            break
          n *= 2
    self._edge_n[eidx] = n
    return n

  def edge_uv(self, eidx, rev, loc_edge, face, surf, loc_surf_inv, n_degenerate):
    """-> (rows (k, 5): u, v, x, y, z;  edge id or -1;  sample parameters (k,)) in wire direction"""
    e = self.P.tshapes[eidx]
    n = self.edge_segments(eidx)
    if n:
      sp = self._edge_s.setdefault(eidx, np.linspace(0.0, 1.0, n + 1))
    else:
      sp = None                                         # degenerated edge: belongs to this face alone
    # BRep_Tool::CurveOnSurface: the representation on this surface whose location is
    # (edge location)^-1 * (surface location) -- one TEdge can bound the same surface twice
    # (the rim circle of a cylinder is one edge, placed at both ends)
    want = np.linalg.inv(loc_edge) @ np.linalg.inv(loc_surf_inv)
    rep, best = None, np.inf
    for r in e.pcurves:
      if r[2] == face.surface:
        d = np.abs(self.P.locations[r[3]] - want).max()
        if d < best:
          rep, best = r, d
    # the 3-D curve gives the vertex positions (shared with the face on the other side: stored
    # p-curves of imported shapes follow it to ~1e-2 mm only), the p-curve the place in (u, v)
    if sp is None:
      nd = n_degenerate
      if isinstance(nd, tuple):
        if rep is None:
          raise BRepError('degenerated edge without a p-curve')
        ends = self.P.curves2d[rep[0] - 1].eval(np.array([rep[4], rep[5]]))
        nd = nd[0] if abs(ends[1, 0] - ends[0, 0]) * nd[0] >= abs(ends[1, 1] - ends[0, 1]) * nd[1] else nd[1]
      sp = np.linspace(0.0, 1.0, nd + 1)
    xyz = np.full((len(sp), 3), np.nan)
    if e.curve3d is not None and not e.degenerated:
      c = self.P.curves[e.curve3d[0] - 1]
      xyz = _xf(loc_edge @ self.P.locations[e.curve3d[1]], c.eval(e.curve3d[2] + sp * (e.curve3d[3] - e.curve3d[2])))
      # the ends are the edge's vertices (curves of imported shapes end within the vertex tolerance only)
      for o_v, vidx, l_v in e.subs:
        v = self.P.tshapes[vidx]
        if v.kind == 'Ve' and o_v in '+-':
          xyz[0 if o_v == '+' else -1] = _xf(loc_edge @ self.P.locations[l_v], v.point[None, :])[0]
    if rep is not None:
      pc_idx = rep[1] if (rev and rep[1] is not None) else rep[0]
      pc = self.P.curves2d[pc_idx - 1]
      uv = pc.eval(rep[4] + sp * (rep[5] - rep[4]))
      if isinstance(surf, (brep.Revolved, brep.Paraboloid)) and np.isfinite(xyz).all():
        # quadrics and tori: the parameters of the 3-D points themselves (stored p-curves of
        # imports are approximations: a rim circle wobbles about its v = const line, and the
        # triangulation fills the bulges with facets that stand across the surface); the
        # p-curve decides the period and stands in at poles
        uv = surf.invert(_xf(loc_surf_inv, xyz), uv)
    elif surf.kind == 'plane' and e.curve3d is not None:
      u, v = surf.invert(_xf(loc_surf_inv, xyz))
      uv = np.stack([u, v], axis=1)
    else:
      raise BRepError(f'edge without a p-curve on a {surf.kind} face')
    out = np.hstack([uv, xyz])
    owner = eidx if n else -1
    return (out[::-1], owner, sp[::-1]) if rev else (out, owner, sp)

  # -- faces -----------------------------------------------------------------
  def grid_steps(self, surf, lo, hi):
    if surf.kind == 'plane':
      return np.inf, np.inf
    if surf.kind == 'bspline-surface':
      # measured: halve the step until the chords of a few iso-parametric lines are within the deflection
      steps = []
      for axis in (0, 1):
        n = 4
        while n < self.max_grid:
          a = np.linspace(lo[axis], hi[axis], n + 1)
          mid = 0.5 * (a[1:] + a[:-1])
          dev = 0.0
          for o in np.linspace(lo[1 - axis], hi[1 - axis], 5):
            oa, om = np.full(len(a), o), np.full(len(mid), o)
            ends = surf.eval(a, oa) if axis == 0 else surf.eval(oa, a)
            mids = surf.eval(mid, om) if axis == 0 else surf.eval(om, mid)
            dev = max(dev, np.linalg.norm(mids - 0.5 * (ends[1:] + ends[:-1]), axis=1).max())
          if dev < self.tol:
            break
          n *= 2
        steps.append((hi[axis] - lo[axis]) / n)
      return steps[0], steps[1]
    if surf.kind == 'paraboloid':
      return surf.steps_range(self.tol, lo, hi)
    du, dv = surf.steps(self.tol)
    if du is None:       # cone: the largest radius of the face's v range
      rmax = max(abs(surf.r + lo[1] * np.sin(surf.extra)), abs(surf.r + hi[1] * np.sin(surf.extra)), 1e-9)
      du = 2.0 * np.arccos(max(-1.0, 1.0 - min(self.tol / rmax, 1.0)))
    return du, dv

  def face(self, fidx, loc_face, reversed_):
    P = self.P
    f = P.tshapes[fidx]
    surf = P.surfaces[f.surface - 1]
    loc_surf = loc_face @ P.locations[f.surface_loc]
    loc_surf_inv = np.linalg.inv(loc_surf)
    # first pass with coarse degenerate edges to learn the face's (u, v) extent, then the real one
    loops, _ = self._loops(f, loc_face, surf, loc_surf_inv, 8)
    allp = np.concatenate(loops)[:, :2]
    lo, hi = allp.min(axis=0), allp.max(axis=0)
    du, dv = self.grid_steps(surf, lo, hi)
    du = max(du, (hi[0] - lo[0]) / self.max_grid) if np.isfinite(du) else du
    dv = max(dv, (hi[1] - lo[1]) / self.max_grid) if np.isfinite(dv) else dv
    # a degenerated edge (pole of a sphere, apex of a surface of revolution) is sampled where the
    # grid lines of the direction it runs along meet it
    n_deg = (max(1, int(np.ceil((hi[0] - lo[0]) / du))) if np.isfinite(du) else 8,
             max(1, int(np.ceil((hi[1] - lo[1]) / dv))) if np.isfinite(dv) else 8)
    loops, segs = self._loops(f, loc_face, surf, loc_surf_inv, n_deg)
    # metric of the parameter plane (for well-shaped triangles in space)
    mid = 0.5 * (lo + hi)
    h = 1e-4 * np.maximum(hi - lo, 1e-9)
    c = surf.eval(np.array([mid[0] - h[0], mid[0] + h[0], mid[0], mid[0]]),
                  np.array([mid[1], mid[1], mid[1] - h[1], mid[1] + h[1]]))
    su = max(np.linalg.norm(c[1] - c[0]) / (2 * h[0]), 1e-9)
    sv = max(np.linalg.norm(c[3] - c[2]) / (2 * h[1]), 1e-9)
    scale = np.array([su, sv])
    if np.isfinite(du) and not np.isfinite(dv):
      # curved along u only (cylinder, cone): the chord error of a facet is set by its extent in u
      # alone, so the triangulation is made in a metric that shrinks v -- facets then run along
      # the straight direction and join neighbouring points of the two rims instead of fanning
      # out from the points of a subdivided seam (the face's height becomes two steps of u)
      scale[1] *= min(1.0, 2.0 * du * su / max((hi[1] - lo[1]) * sv, 1e-300))
    interior = self._interior(loops, lo, hi, du, dv, scale)
    pts, tri = _conforming_delaunay(loops, interior, scale, segs, self.requests)
    if len(tri) == 0:
      raise BRepError('a face produced no facets')
    uv = pts[:, :2]
    x = _xf(loc_surf, surf.eval(uv[:, 0], uv[:, 1]))
    known = np.isfinite(pts[:, 2])
    x[known] = pts[known, 2:]
    n = surf.normal(uv[:, 0], uv[:, 1]) @ loc_surf[:3, :3].T
    if reversed_:
      n, tri = -n, tri[:, [0, 2, 1]]
    a, b, c = x[tri[:, 0]], x[tri[:, 1]], x[tri[:, 2]]
    fn = np.cross(b - a, c - a)
    area2 = np.linalg.norm(fn, axis=1)
    keep = area2 > 1e-12 * max(area2.max(), 1e-300)
    # a facet lies along the surface: one that stands across it (three points of one wobbling
    # boundary line) is not part of the face
    along = np.einsum('ij,ij->i', fn, n[tri[:, 0]] + n[tri[:, 1]] + n[tri[:, 2]])
    keep &= along > 0.5 * 3 * area2
    return x, n, tri[keep], surf.kind, float(0.5 * area2[keep].sum())

  def _loops(self, f, loc_face, surf, loc_surf_inv, n_degenerate):
    P = self.P
    loops, segs = [], []
    for o_w, widx, l_w in f.subs:
      w = P.tshapes[widx]
      if w.kind != 'Wi' or o_w in 'ie':
        continue
      loc_w = loc_face @ P.locations[l_w]
      wire_rev = o_w == '-'
      parts = []
      for o_e, eidx, l_e in w.subs:
        if o_e in 'ie' or P.tshapes[eidx].kind != 'Ed':
          continue
        rev = (o_e == '-') ^ wire_rev
        parts.append(self.edge_uv(eidx, rev, loc_w @ P.locations[l_e], f, surf, loc_surf_inv, n_degenerate))
      if wire_rev:
        parts = parts[::-1]
      if parts:
        loop, seg = _chain(parts)
        loops.append(loop)
        segs.append(seg)
    if not loops:
      raise BRepError('a face without wires (natural bounds) is not meshed')
    return loops, segs

  def _interior(self, loops, lo, hi, du, dv, scale):
    if not (np.isfinite(du) or np.isfinite(dv)):
      return np.zeros((0, 2))
    def axis(a, b, d):
      if not np.isfinite(d):
        return None
      k = max(1, int(np.ceil((b - a) / d)))
      return a + (b - a) * np.arange(1, k) / k
    us, vs = axis(lo[0], hi[0], du), axis(lo[1], hi[1], dv)
    if us is None or vs is None:
      # curved in one direction only (cylinder, cone): facets span the straight direction
      return np.zeros((0, 2))
    if len(us) == 0 or len(vs) == 0:
      return np.zeros((0, 2))
    g = np.stack(np.meshgrid(us, vs, indexing='ij'), axis=-1).reshape(-1, 2)
    g = g[_inside(g, loops)]
    if len(g) == 0:
      return g
    b = np.concatenate(loops)[:, :2] * scale
    d, _ = cKDTree(b).query(g * scale)
    return g[d > 0.6 * min(du * scale[0], dv * scale[1])]


def _chain(parts):
  """edge polylines of one wire -> one closed polyline (stored order, else nearest start) and, per
  chord, the edge it belongs to and the edge parameter of its mid-point: (k, 2) array"""
  todo = list(parts)
  out = [todo.pop(0)]
  while todo:
    end = out[-1][0][-1]
    k = min(range(len(todo)), key=lambda i: np.abs(todo[i][0][0, :2] - end[:2]).sum())
    out.append(todo.pop(k))
  pts = [out[0][0]] + [rows[1:] for rows, _, _ in out[1:]]
  seg = [np.stack([np.full(len(sp) - 1, float(owner)), 0.5 * (sp[1:] + sp[:-1])], axis=1) for _, owner, sp in out]
  loop, seg = np.concatenate(pts), np.concatenate(seg)
  if np.abs(loop[0, :2] - loop[-1, :2]).sum() > 1e-6 * max(1.0, np.abs(loop[:, :2]).max()):
    loop = np.concatenate([loop, loop[:1]])
    seg = np.concatenate([seg, [[-1.0, 0.0]]])
  else:
    loop[-1] = loop[0]       # exactly closed: the even-odd test counts crossings of closed wires
  return loop, seg


def _inside(pts, loops):
  """even-odd rule against all wires of a face"""
  seg_a = np.concatenate([l[:-1, :2] for l in loops])
  seg_b = np.concatenate([l[1:, :2] for l in loops])
  out = np.zeros(len(pts), dtype=bool)
  for s in range(0, len(pts), 4096):
    p = pts[s:s + 4096]
    ya, yb = seg_a[None, :, 1], seg_b[None, :, 1]
    py, px = p[:, 1:2], p[:, 0:1]
    cond = (ya > py) != (yb > py)
    with np.errstate(divide='ignore', invalid='ignore'):
      xi = seg_a[None, :, 0] + (py - ya) * (seg_b[None, :, 0] - seg_a[None, :, 0]) / (yb - ya)
    out[s:s + 4096] = (np.count_nonzero(cond & (px < xi), axis=1) % 2) == 1
  return out


def _enclosed_area(loops):
  """area enclosed by the wires under the even-odd rule (wires of a face do not cross:
  outer boundaries minus holes, whatever their stored sense)"""
  areas = []
  for l in loops:
    x, y = l[:, 0], l[:, 1]
    areas.append(0.5 * abs(np.dot(x[:-1], y[1:]) - np.dot(x[1:], y[:-1])))
  total = 0.0
  for k, l in enumerate(loops):
    depth = sum(1 for j, m in enumerate(loops) if j != k and _inside(l[:1, :2], [m])[0])
    total += areas[k] if depth % 2 == 0 else -areas[k]
  return total


def _conforming_delaunay(loops, interior, scale, segs=None, requests=None):
  loops = [l.copy() for l in loops]
  segs = [np.array(g) for g in segs] if segs is not None else [np.full((len(l) - 1, 2), -1.0) for l in loops]
  for _ in range(12):
    inner = np.hstack([interior, np.full((len(interior), 3), np.nan)])
    pts = np.concatenate([l[:-1] for l in loops] + [inner])
    uniq, inv = np.unique(np.round(pts[:, :2] * scale / 1e-9).astype(np.int64), axis=0, return_inverse=True)
    inv = inv.ravel()
    first = np.full(len(uniq), -1, dtype=np.int64)
    first[inv[::-1]] = np.arange(len(pts))[::-1]
    p = pts[first]
    if len(p) < 3:
      return p, np.zeros((0, 3), dtype=np.int64)
    tri = Delaunay(p[:, :2] * scale, qhull_options='Qbb Qc Qz Q12 Qt').simplices
    edges = set()
    for a, b in ((0, 1), (1, 2), (2, 0)):
      e = np.sort(tri[:, [a, b]], axis=1)
      edges.update(map(tuple, e))
    missing, off = False, 0
    new_loops, new_segs = [], []
    for l, g in zip(loops, segs):
      n = len(l) - 1
      ids = inv[off:off + n]
      off += n
      nxt = np.roll(ids, -1)
      out, gout = [], []
      for k in range(n):
        out.append(l[k])
        a, b = int(ids[k]), int(nxt[k])
        if a != b and (min(a, b), max(a, b)) not in edges:
          out.append(0.5 * (l[k] + l[k + 1]))      # (u, v) and the chord's own mid-point in space
          if g[k, 0] >= 0 and requests is not None:
            requests.append((int(g[k, 0]), float(g[k, 1])))
          gout += [[-1.0, 0.0], [-1.0, 0.0]]       # halves of a split chord are this face's own
          missing = True
        else:
          gout.append(g[k])
      out.append(l[-1])
      new_loops.append(np.array(out))
      new_segs.append(np.array(gout).reshape(-1, 2))
    if not missing:
      break
    loops, segs = new_loops, new_segs
  # needles (three almost collinear boundary points): their centroid lies on the wire, the
  # even-odd test cannot place them, and they cover nothing
  q = p[:, :2] * scale
  a, b, c = q[tri[:, 0]], q[tri[:, 1]], q[tri[:, 2]]
  area2 = np.abs((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))
  longest = np.maximum(np.maximum(((b - a) ** 2).sum(1), ((c - b) ** 2).sum(1)), ((a - c) ** 2).sum(1))
  tri = tri[area2 > 1e-7 * longest]
  cen = p[tri][:, :, :2].mean(axis=1)
  tri = tri[_inside(cen, loops)]
  # the facets must tile the region the wires enclose
  a, b, c = p[tri[:, 0], :2], p[tri[:, 1], :2], p[tri[:, 2], :2]
  got = 0.5 * np.abs((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])).sum()
  want = _enclosed_area(loops)
  if abs(got - want) > 1e-4 * max(want, 1e-300):
    raise BRepError(f'triangulation of a face covers {got:.9g} of {want:.9g} in its parameter plane')
  # counter-clockwise in (u, v)
  a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
  cw = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])) < 0
  tri[cw] = tri[cw][:, [0, 2, 1]]
  return p, tri


def tessellate(payload, deflection=1e-3, max_grid=1024, keep_root_location=True):
  """-> ShapeMesh of every face of the payload (explorer order = FreeCAD's Face1, Face2, ...).
  deflection: largest distance between a facet and the surface (mm);
  keep_root_location=False leaves out the location stored with the root shape (FreeCAD keeps the
  object's Placement there, property Shape of Part::Feature)."""
  if not isinstance(payload, brep.Payload):
    payload = brep.load(payload)
  m = _Mesher(payload, deflection, max_grid)
  base = np.eye(4) if keep_root_location else np.linalg.inv(payload.locations[payload.root[2]])
  # a chord one face has to split is split for the face on its other side too: the edge's sample
  # list is refined and all faces are meshed again (the last pass keeps what is left as chord
  # mid-points, which lie on the neighbour's facet edge)
  for _ in range(8):
    m.requests = []
    vs, ns, ts, faces = [], [], [], []
    nv = nt = 0
    for k, (fidx, loc, rev) in enumerate(payload.faces(), start=1):
      x, n, tri, kind, area = m.face(fidx, base @ loc, rev)
      vs.append(x)
      ns.append(n)
      ts.append(tri + nv)
      faces.append(FaceMesh(k, kind, nt, len(tri), area))
      nv += len(x)
      nt += len(tri)
    if not m.requests:
      break
    for eidx, sm in m.requests:
      m._edge_s[eidx] = np.unique(np.append(m._edge_s[eidx], sm))
  v, n = np.concatenate(vs), np.concatenate(ns)
  n = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-300)
  return ShapeMesh(v, n, np.concatenate(ts), faces)
