"""Fan mode of surface sources: rays along the face normals on a grid of roughly
equidistant points (SurfaceSourceProxy._makeSurfaceGrid / _generateRays(mode='fans'),
freecad_elements/surface_source.py:119-268, 467-519).

The reference builds the grid from four facilities of an OpenCASCADE face:
`ParameterRange`, `valueAt(u, v)`, "is this point on the face" (distance to the
face below the distance tolerance) and `derivative1At(u, v)` -- a rectilinear
(u, v) grid over the parameter range, refined in five passes that estimate how
much of the range the face fills, the effective lengths of the two parameter
lines and whether the area element is uniform along one parameter, and then thin
out the rows where parameter lines crowd (poles of a sphere).  `FaceView` offers
the same four facilities for the faces this build knows: faces of stored BRep
shapes (scene/brep.py: surface + trimming wires), the faces of the untrimmed
parametric primitives in OpenCASCADE's parametrisations, and the faces of boolean
results over such primitives (each operand's faces, trimmed by the other operands).  The grid logic is
restated on top of it; rays are handed to the tracer as explicit initial
conditions (`odw_trace_rays`).
"""
import warnings

import numpy as np

from ..scene import geometry


class FaceView:
  """a face in global coordinates: range = (u0, u1, v0, v1), area,
  value(u, v) -> point, valid(u, v, point) -> bool, derivatives(u, v) -> (dS/du, dS/dv), normal(u, v)"""

  def __init__(self, rng, area, value, valid, normal, derivatives):
    self.range, self.area, self.value, self.valid, self.normal = rng, area, value, valid, normal
    # exact derivatives (OpenCASCADE's derivative1At is analytic): the grid logic compares area
    # elements against distTol^2, which difference quotients of ordinary accuracy do not survive
    self.derivatives = derivatives


# ------------------------------------------------------------------ faces of primitives
def _primitive_face_table(kind, params, to_world, dist_tol):
  """{face bit position: FaceView} of an untrimmed primitive in OpenCASCADE's parametrisations"""
  R, t = to_world.m[:3, :3], to_world.m[:3, 3]
  world = lambda p: R @ np.asarray(p, float) + t
  wdir = lambda d: R @ np.asarray(d, float)
  two_pi = 2 * np.pi
  p = params
  faces = {}
  if kind == geometry.BOX:
    for f in range(6):
      a, b1, b2 = f >> 1, ((f >> 1) + 1) % 3, ((f >> 1) + 2) % 3
      n = np.zeros(3)
      n[a] = 1.0 if f & 1 else -1.0

      def value(u, v, a=a, b1=b1, b2=b2, f=f):
        c = np.zeros(3)
        c[a], c[b1], c[b2] = (p[a] if f & 1 else 0.0), u, v
        return world(c)
      e1, e2 = np.zeros(3), np.zeros(3)
      e1[b1] = e2[b2] = 1.0
      faces[f] = FaceView((0.0, p[b1], 0.0, p[b2]), p[b1] * p[b2], value, lambda u, v, x: True,
                          lambda u, v, n=n: wdir(n), lambda u, v, e1=e1, e2=e2: (wdir(e1), wdir(e2)))
  elif kind == geometry.SPHERE:
    r = p[0]
    unit = lambda u, v: np.array([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)])
    faces[0] = FaceView((0.0, two_pi, -np.pi / 2, np.pi / 2), 4 * np.pi * r * r, lambda u, v: world(r * unit(u, v)),
                        lambda u, v, x: True, lambda u, v: wdir(unit(u, v)),
                        lambda u, v: (wdir([-r * np.cos(v) * np.sin(u), r * np.cos(v) * np.cos(u), 0.0]),
                                      wdir([-r * np.sin(v) * np.cos(u), -r * np.sin(v) * np.sin(u), r * np.cos(v)])))
  elif kind == geometry.TORUS:
    r1, r2 = p[0], p[1]
    faces[0] = FaceView(
        (0.0, two_pi, 0.0, two_pi), 4 * np.pi**2 * r1 * r2,
        lambda u, v: world([(r1 + r2 * np.cos(v)) * np.cos(u), (r1 + r2 * np.cos(v)) * np.sin(u), r2 * np.sin(v)]),
        lambda u, v, x: True, lambda u, v: wdir([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)]),
        lambda u, v: (wdir([-(r1 + r2 * np.cos(v)) * np.sin(u), (r1 + r2 * np.cos(v)) * np.cos(u), 0.0]),
                      wdir([-r2 * np.sin(v) * np.cos(u), -r2 * np.sin(v) * np.sin(u), r2 * np.cos(v)])))
  elif kind in (geometry.CYLINDER, geometry.CONE):
    r1, r2, h = (p[0], p[0], p[1]) if kind == geometry.CYLINDER else (p[0], p[1], p[2])
    slant = np.hypot(h, r2 - r1)
    k = (r2 - r1) / h

    def lateral(u, v):
      z = h * v / slant
      r = r1 + k * z
      return world([r * np.cos(u), r * np.sin(u), z])
    inv = 1.0 / np.sqrt(1 + k * k)
    faces[0] = FaceView((0.0, two_pi, 0.0, slant), np.pi * (r1 + r2) * slant, lateral, lambda u, v, x: True,
                        lambda u, v: wdir([np.cos(u) * inv, np.sin(u) * inv, -k * inv]),
                        lambda u, v: (wdir([-(r1 + k * h * v / slant) * np.sin(u), (r1 + k * h * v / slant) * np.cos(u), 0.0]),
                                      wdir([k * h / slant * np.cos(u), k * h / slant * np.sin(u), h / slant])))
    for f, rr, z, s in ((1, r1, 0.0, -1.0), (2, r2, h, 1.0)):
      if rr > 0:
        faces[f] = FaceView((-rr, rr, -rr, rr), np.pi * rr * rr, lambda u, v, z=z: world([u, v, z]),
                            lambda u, v, x, rr=rr: np.hypot(u, v) <= rr + dist_tol,
                            lambda u, v, s=s: wdir([0.0, 0.0, s]),
                            lambda u, v: (wdir([1.0, 0.0, 0.0]), wdir([0.0, 1.0, 0.0])))
  elif kind == geometry.PARABOLOID:
    # surface of revolution of the parabola (v, 0, v^2 / 4f) about z: u = angle, v = distance from the axis
    f, h = p[0], p[1]
    rr = 2.0 * np.sqrt(f * h)
    faces[0] = FaceView(
        (0.0, two_pi, 0.0, rr), np.pi * rr / (6 * h * h) * ((rr * rr + 4 * h * h)**1.5 - rr**3),
        lambda u, v: world([v * np.cos(u), v * np.sin(u), v * v / (4 * f)]), lambda u, v, x: True,
        lambda u, v: wdir(np.array([v * np.cos(u), v * np.sin(u), -2 * f]) / np.sqrt(v * v + 4 * f * f)),
        lambda u, v: (wdir([-v * np.sin(u), v * np.cos(u), 0.0]), wdir([np.cos(u), np.sin(u), v / (2 * f)])))
    faces[2] = FaceView((-rr, rr, -rr, rr), np.pi * rr * rr, lambda u, v: world([u, v, h]),
                        lambda u, v, x: np.hypot(u, v) <= rr + dist_tol, lambda u, v: wdir([0.0, 0.0, 1.0]),
                        lambda u, v: (wdir([1.0, 0.0, 0.0]), wdir([0.0, 1.0, 0.0])))
  else:
    raise geometry.UnsupportedGeometry(f'no fan grid for primitive kind {kind}')
  return faces


def _primitive_faces(kind, params, to_world, names, dist_tol):
  """FaceViews of an untrimmed primitive; names: 'Face<k>' or all"""
  from . import surface_source
  faces = _primitive_face_table(kind, params, to_world, dist_tol)
  if kind == geometry.PARABOLOID:
    if names:
      raise geometry.UnsupportedGeometry('faces of a paraboloid are not selected by name')
    return [faces[0], faces[2]]
  if names:
    return [faces[surface_source._faceIndex(kind, params, n)] for n in names]
  # whole body: OpenCASCADE's face order (surface_source._faceIndex)
  out, k = [], 1
  while True:
    try:
      out.append(faces[surface_source._faceIndex(kind, params, f'Face{k}')])
    except geometry.UnsupportedGeometry:
      return out
    k += 1


# ------------------------------------------------------------------ faces of boolean results
def _inside_primitive(fp, x, tol):
  """is the point inside the (closed) primitive, or within tol of its boundary"""
  m = fp.to_world.m
  q = (np.asarray(x, float) - m[:3, 3]) @ m[:3, :3]
  p, k = fp.params, fp.kind
  if k == geometry.BOX:
    return bool(np.all(q >= -tol) and np.all(q <= np.array(p[:3]) + tol))
  if k == geometry.SPHERE:
    return bool(np.linalg.norm(q) <= p[0] + tol)
  rho = np.hypot(q[0], q[1])
  if k == geometry.CYLINDER:
    return bool(rho <= p[0] + tol and -tol <= q[2] <= p[1] + tol)
  if k == geometry.CONE:
    z = min(max(q[2], 0.0), p[2])
    return bool(-tol <= q[2] <= p[2] + tol and rho <= p[0] + (p[1] - p[0]) * z / p[2] + tol * np.hypot(1.0, (p[1] - p[0]) / p[2]))
  if k == geometry.TORUS:
    return bool(np.hypot(rho - p[0], q[2]) <= p[1] + tol)
  if k == geometry.PARABOLOID:
    return bool(q[2] <= p[1] + tol and rho * rho <= 4 * p[0] * q[2] + tol * np.sqrt(rho * rho + 4 * p[0] * p[0]) * 2)
  raise geometry.UnsupportedGeometry(f'no containment test for primitive kind {k}')


def _strictly_inside_primitive(fp, x, tol):
  m = fp.to_world.m
  q = (np.asarray(x, float) - m[:3, 3]) @ m[:3, :3]
  p, k = fp.params, fp.kind
  if k == geometry.BOX:
    return bool(np.all(q > tol) and np.all(q < np.array(p[:3]) - tol))
  if k == geometry.SPHERE:
    return bool(np.linalg.norm(q) < p[0] - tol)
  rho = np.hypot(q[0], q[1])
  if k == geometry.CYLINDER:
    return bool(rho < p[0] - tol and tol < q[2] < p[1] - tol)
  if k == geometry.CONE:
    return bool(tol < q[2] < p[2] - tol and rho < p[0] + (p[1] - p[0]) * q[2] / p[2] - tol * np.hypot(1.0, (p[1] - p[0]) / p[2]))
  if k == geometry.TORUS:
    return bool(np.hypot(rho - p[0], q[2]) < p[1] - tol)
  if k == geometry.PARABOLOID:
    return bool(q[2] < p[1] - tol and rho * rho < 4 * p[0] * q[2] - tol * np.sqrt(rho * rho + 4 * p[0] * p[0]) * 2)
  raise geometry.UnsupportedGeometry(f'no containment test for primitive kind {k}')


_TRIM_SCAN = 96       # samples per parameter when the trimmed extent of a face is looked for


def _boolean_faces(tree, container, dist_tol):
  """FaceViews of the result of Cut / Fuse / Common over primitives: every face of every operand that keeps a part
  of itself in the result, trimmed by the other operands.  OpenCASCADE hands the reference such a result as faces
  with the trimmed `ParameterRange` and an `isInside` that knows the trimming wires; here the trimming is the
  conjunction of the operands' half-space conditions (geometry.flatten, the same conditions the tracer trims
  with), the parameter range is the extent of the part that survives (scanned, then each bound refined by
  bisection) and the area the sum of the area elements over the scan.  One view per operand face: OpenCASCADE
  splits a face whose remainder is not connected, the grid recipe then runs per piece -- here it runs once over
  their common extent.  Order: operands depth-first, faces in the primitive's own order."""
  out = []
  for fp in geometry.flatten(tree, container):
    table = _primitive_face_table(fp.kind, fp.params, fp.to_world, dist_tol)
    for f in sorted(table):
      if not (fp.facemask >> f) & 1:
        continue
      view = table[f]

      def valid(u, v, x, view=view, conds=fp.conds):
        if not view.valid(u, v, x):
          return False
        for other, inside in conds:
          # (a point of the face that lies ON the other operand's boundary belongs to the result's edge)
          if inside and not _inside_primitive(other, x, dist_tol):
            return False
          if not inside and _strictly_inside_primitive(other, x, dist_tol):
            return False
        return True
      u0, u1, v0, v1 = view.range
      found = True
      for _ in range(4):
        # scan; where the surviving part fills less than half of the window in a direction, look again closer
        us, vs = np.linspace(u0, u1, _TRIM_SCAN + 1), np.linspace(v0, v1, _TRIM_SCAN + 1)
        ok = np.array([[valid(u, v, view.value(u, v)) for v in vs] for u in us])
        if not ok.any():
          found = False
          break
        iu, iv = np.nonzero(ok.any(axis=1))[0], np.nonzero(ok.any(axis=0))[0]
        if max(iu[-1] - iu[0], 1) * 2 > _TRIM_SCAN and max(iv[-1] - iv[0], 1) * 2 > _TRIM_SCAN:
          break
        u0, u1 = us[max(iu[0] - 1, 0)], us[min(iu[-1] + 1, _TRIM_SCAN)]
        v0, v1 = vs[max(iv[0] - 1, 0)], vs[min(iv[-1] + 1, _TRIM_SCAN)]
      if not found:
        continue

      def any_on(axis, c):
        if axis == 0:
          return any(valid(c, v, view.value(c, v)) for v in vs)
        return any(valid(u, c, view.value(u, c)) for u in us)

      def refine(axis, grid, k_in, k_out):
        """between sample k_in (part of the result somewhere along it) and its neighbour k_out (none)"""
        if not 0 <= k_out < len(grid):
          return grid[k_in]
        a, b = grid[k_in], grid[k_out]
        for _ in range(24):
          c = 0.5 * (a + b)
          if any_on(axis, c):
            a = c
          else:
            b = c
        return a
      rng = (refine(0, us, iu[0], iu[0] - 1), refine(0, us, iu[-1], iu[-1] + 1),
             refine(1, vs, iv[0], iv[0] - 1), refine(1, vs, iv[-1], iv[-1] + 1))
      # area of what survives: area elements at the cell centres of the last scan
      uc, vc = 0.5 * (us[1:] + us[:-1]), 0.5 * (vs[1:] + vs[:-1])
      area = 0.0
      for u in uc:
        for v in vc:
          if valid(u, v, view.value(u, v)):
            du, dv = view.derivatives(u, v)
            area += float(np.linalg.norm(np.cross(du, dv)))
      area *= (us[1] - us[0]) * (vs[1] - vs[0])
      if area <= 0:
        continue
      normal = (lambda u, v, view=view: -np.asarray(view.normal(u, v))) if fp.flip else view.normal
      out.append(FaceView(rng, area, view.value, valid, normal, view.derivatives))
  return out


# ------------------------------------------------------------------ faces of stored shapes
def _brep_faces(payload, name, to_world, names, dist_tol):
  from ..scene import brep, brep_mesh
  from ..scene.brep_mesh import _inside, _xf
  P = brep.load(payload.data)
  base = np.linalg.inv(P.locations[P.root[2]])        # the stored root location is the object's Placement
  M = to_world.m
  mesh = brep_mesh.tessellate(P, deflection=geometry.BREP_DEFLECTION, keep_root_location=False)
  mesher = brep_mesh._Mesher(P, 1e-4 * max(float(np.ptp(mesh.vertices, axis=0).max()), 1e-9), 256)
  all_faces = P.faces()
  if names:
    picks = []
    for n in names:
      if not n.startswith('Face') or not 1 <= int(n[4:]) <= len(all_faces):
        raise geometry.UnsupportedGeometry(f'{name} has no sub-element {n!r} ({len(all_faces)} faces)')
      picks.append(int(n[4:]) - 1)
  else:
    picks = range(len(all_faces))
  out = []
  for k in picks:
    fidx, loc, rev = all_faces[k]
    f = P.tshapes[fidx]
    surf = P.surfaces[f.surface - 1]
    local = base @ loc @ P.locations[f.surface_loc]
    loc_surf = M @ local
    loops, _ = mesher._loops(f, base @ loc, surf, np.linalg.inv(local), 64)
    uv = np.concatenate(loops)[:, :2]
    edge_pts = np.concatenate([_xf(loc_surf, surf.eval(l[:, 0], l[:, 1])) for l in loops])

    def value(u, v, surf=surf, loc_surf=loc_surf):
      return _xf(loc_surf, surf.eval(np.array([u]), np.array([v])))[0]

    def valid(u, v, x, loops=loops, edge_pts=edge_pts):
      if _inside(np.array([[u, v]]), loops)[0]:
        return True
      return bool(np.linalg.norm(edge_pts - x, axis=1).min() < max(dist_tol, 1e-6))   # on the rim

    def normal(u, v, surf=surf, loc_surf=loc_surf, rev=rev):
      n = loc_surf[:3, :3] @ surf.normal(np.array([u]), np.array([v]))[0]
      n = n / np.linalg.norm(n)
      return -n if rev else n
    def derivatives(u, v, surf=surf, loc_surf=loc_surf):
      du, dv = surf.d1(np.array([u]), np.array([v]))
      return loc_surf[:3, :3] @ du[0], loc_surf[:3, :3] @ dv[0]
    out.append(FaceView((uv[:, 0].min(), uv[:, 0].max(), uv[:, 1].min(), uv[:, 1].max()), mesh.faces[k].area,
                        value, valid, normal, derivatives))
  return out


def facesOf(doc, source):
  """[FaceView] of a surface source's ActiveSurfaces, one set per global placement of each part
  (surface_source.py:436-459)"""
  from . import surface_source
  from ..scene import bake as _bake
  tol = surface_source.distTol(doc)
  out = []
  for part, subs in source._props.get('ActiveSurfaces') or []:
    own = part.Placement if part.hasProperty('Placement') else None
    names = [s for s in subs if s]
    for gp in _bake.globalPlacements(doc, part):
      container = gp * own.inverse() if own is not None else gp     # solids_of() applies part.Placement itself
      for tree in geometry.solids_of(part, brepFacets=True):
        pl = container * tree.placement
        if tree.op == 'prim':
          out.extend(_primitive_faces(tree.kind, tree.params, pl, names, tol))
        elif tree.op == 'mesh' and len(tree.mesh) > 4:
          out.extend(_brep_faces(tree.mesh[4], tree.source, pl, names, tol))
        elif tree.op in ('cut', 'fuse', 'common'):
          if names:
            raise geometry.UnsupportedGeometry(
                f'{source.Name}: faces {names} of the boolean result {part.Name} are numbered by OpenCASCADE; '
                f'select the whole body or faces of primitive solids')
          out.extend(_boolean_faces(tree, container, tol))
        else:
          raise NotImplementedError(
              f'{source.Name}: fan grids are built on faces of primitives, boolean results and stored shapes; '
              f'{tree.source or part.Name} is a {tree.op}')
  return out


# ------------------------------------------------------------------ the grid
def makeSurfaceGrid(face, totalGridPoints, distTol):
  """-> [((u, v), point, (dS/du, dS/dv))]: roughly equidistant points on the face
  (_makeSurfaceGrid, surface_source.py:119-267: five passes, see the module docstring)"""
  limits = dict(u=(face.range[0], face.range[1]), v=(face.range[2], face.range[3]))
  sizes = dict(u=face.range[1] - face.range[0], v=face.range[3] - face.range[2])
  uniform, fill, eff = None, 1.0, sizes
  for depth in range(5):
    order = 'uv' if eff['u'] >= eff['v'] else 'vu'
    if uniform is not None:
      order = uniform + ('v' if uniform == 'u' else 'u')
    axes = []
    for p, q in zip(order, order[::-1]):
      n = max(5, 1 + int(2 * np.round(np.sqrt(eff[p] / eff[q] * totalGridPoints / fill) / 2)))
      axes.append(np.linspace(limits[p][0], limits[p][1], n))
    P1, P2 = axes
    s1, s2 = P1[1] - P1[0], P2[1] - P2[0]
    uv = (lambda a, b: (a, b)) if order == 'uv' else (lambda a, b: (b, a))
    pts = [[face.value(*uv(a, b)) for b in P2] for a in P1]
    ok = [[bool(face.valid(*uv(a, b), pts[i][j])) for j, b in enumerate(P2)] for i, a in enumerate(P1)]
    der = [[uv(*face.derivatives(*uv(a, b))) if ok[i][j] else (None, None) for j, b in enumerate(P2)]
           for i, a in enumerate(P1)]
    d1 = [[None if d[0] is None else float(np.linalg.norm(d[0])) for d in row] for row in der]
    d2 = [[None if d[1] is None else float(np.linalg.norm(d[1])) for d in row] for row in der]
    area = [[None if a is None else a * b for a, b in zip(r1, r2)] for r1, r2 in zip(d1, d2)]
    known = lambda seq: [x for x in seq if x is not None]

    def is_uniform(rows):
      for row in rows:
        vals = known(row)
        if vals:
          avg = float(np.mean(vals))
          if any(abs(x - avg) * s1 * s2 >= distTol**2 for x in vals):
            return False
      return True
    if is_uniform(area):
      uniform = order[0]
    elif is_uniform(list(zip(*area))):
      uniform = order[1]
    else:
      uniform = None
    len1 = sum(max(known(row)) for row in d1 if known(row)) * s1
    len2 = sum(max(known(col)) for col in zip(*d2) if known(col)) * s2
    eff = {order[0]: len1, order[1]: len2}
    if uniform is not None:
      # rows whose second parameter line is short (towards a pole) keep every 2^k-th point
      for i, row in enumerate(d2):
        row_len = max(1e-20, s2 * sum(known(row)))
        keep = 2.0 ** np.round(np.log2(len2 / row_len)) if len2 > 0 else 1.0
        if keep > len(ok[i]):
          for j in range(1, len(ok[i])):
            ok[i][j] = False
        else:
          for j in range(len(ok[i])):
            if j % keep != 0:
              ok[i][j] = False
    count = sum(sum(row) for row in ok)
    fill = max(fill / 10, count / (len(P1) * len(P2)))
  # very small requests: drop every other row / column of what the minimum 5 x 5 grid produced
  drops = [lambda i, j: False, lambda i, j: i % 2 == 0 or j % 2 == 0,
           lambda i, j: ((i + 1) // 2) % 2 == 0 or ((j + 1) // 2) % 2 == 0]
  while drops and totalGridPoints < 20 and count > totalGridPoints:
    drop = drops.pop(0)
    ok = [[ok[i][j] and not drop(i, j) for j in range(len(P2))] for i in range(len(P1))]
    count = sum(sum(row) for row in ok)
  return [(uv(a, b), pts[i][j], uv(*der[i][j])) for i, a in enumerate(P1) for j, b in enumerate(P2) if ok[i][j]]


def _custom_round(x):
  """ray counts per face: 1, 4, 9 or any integer above (surface_source.py:476)"""
  return int(np.round(x)) if x > 9 else [1, 4, 9][int(np.argmin(np.abs(x - np.array([1, 4, 9]))))]


def generateFanRays(doc, source):
  """-> [(origin, direction, metadata)] of `_generateRays(mode='fans')` (surface_source.py:467-519):
  faces share FanModeRayCount by area, each gets a grid of normal rays"""
  from . import surface_source
  tol = surface_source.distTol(doc)
  faces = facesOf(doc, source)
  if not faces:
    warnings.warn(f'surface source {source.Name} has no ActiveFaces selected for emission')
    return []
  areas = np.array([f.area for f in faces], dtype=np.float64)
  weights = areas / areas.sum()
  total = float(source._props.get('FanModeRayCount', 100))
  wanted = sum(_custom_round(w * total) for w in weights)
  skip = max(0.0, 1 - total / wanted)
  if skip > 0.3:
    warnings.warn(f'cannot place rays on all surfaces, because this would require {wanted} rays, which exceeds '
                  f'FanModeRayCount={total:g}. Skipping {1e2 * skip:.0f}% of faces.')
  else:
    skip = 0.0
  rays, face_i = [], 0.0
  for w, face in zip(weights, faces):
    if skip > 0:
      step = skip / w * len(faces)
      if np.round(face_i) != np.round(face_i + step):
        continue
      face_i += step
    for (u, v), point, (du, dv) in makeSurfaceGrid(face, _custom_round(w * total), tol):
      n = face.normal(u, v)
      rays.append((np.asarray(point, float), n / np.linalg.norm(n), dict(initPhi=0.0, initTheta=0.0)))
  return rays
