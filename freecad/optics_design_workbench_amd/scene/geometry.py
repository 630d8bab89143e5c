"""Parametric solids of a FreeCAD document as CSG trees over analytic primitives.

The reference intersects rays with the *analytic* trimmed surfaces FreeCAD's
`Part` module exposes (`cachedShells/cachedFaces/cachedSurface`,
simulation/raytracing_cache.py:92-111, used at freecad_elements/ray.py:345-411).
Without FreeCAD the same faces are rebuilt from the parametric features:
every face of a boolean result is a face of one operand restricted to the
inside (Common), outside (Fuse, Cut base) or inside-with-flipped-normal (Cut
tool) of the other operands.
"""
from dataclasses import dataclass, field

import numpy as np

from .placement import Placement

BOX, SPHERE, CYLINDER, CONE, TORUS, TRIANGLE, PARABOLOID = range(7)
KIND_NAMES = ['box', 'sphere', 'cylinder', 'cone', 'torus', 'triangle', 'paraboloid']
# (paraboloid: face 0 = the surface of revolution, face 2 = the cap at z = H as on cylinders and
#  cones; there is no face 1)
N_FACES = {BOX: 6, SPHERE: 1, CYLINDER: 3, CONE: 3, TORUS: 1, TRIANGLE: 1, PARABOLOID: 3}
PARABOLOID_FACES = 0b101


class UnsupportedGeometry(ValueError):
  """scene feature that needs FreeCAD/OpenCASCADE to evaluate (BRep import,
  partial revolutions, free-form surfaces)"""


@dataclass
class Node:
  op: str                       # 'prim' | 'common' | 'cut' | 'fuse' | 'mesh'
  placement: Placement = field(default_factory=Placement.identity)
  mesh: tuple = None            # op 'mesh': (vertices (n,3), triangles (m,3) int, vertex normals (n,3) or None)
  kind: int = -1
  params: tuple = ()
  children: list = field(default_factory=list)
  source: str = ''              # document object name
  facemask: int = -1            # op 'prim': faces that exist (-1: all); auxiliary primitives of
                                # recognised BRep solids (scene/brep_csg.py) contribute one face


def _close(a, b, tol=1e-9):
  return abs(float(a) - float(b)) <= tol


def _half_space(theta_deg, size, source):
  """the points whose azimuth about the local z axis lies in [theta, theta + 180 deg], as a box large enough to stand for
  the half-space (its face through the axis is the cutting plane; the others lie outside anything of extent `size`)"""
  L = 4.0 * float(size)
  pl = Placement(base=(0.0, 0.0, 0.0), quat=tuple(np.r_[np.array([0.0, 0.0, 1.0]) * np.sin(np.radians(theta_deg) / 2),
                                                        np.cos(np.radians(theta_deg) / 2)]))
  return Node('prim', placement=pl * Placement(base=(-L, 0.0, -L)), kind=BOX, params=(2 * L, L, 2 * L, 0.0), source=source)


def _swept(prim, angle_deg, size, name):
  """`prim` (a solid of revolution about its local z axis) cut down to the azimuths [0, angle]: what OpenCASCADE builds
  for a partial revolution (BRepPrim_OneAxis: the meridian swept counter-clockwise from the local x axis, closed by two
  planar faces through the axis).  Up to half a turn that is the common part with two half-spaces -- a conjunction, which
  the trimming lists of the flat scene hold; beyond half a turn it is a disjunction (inside one half-space OR inside
  the other), which they do not: such solids need the shape FreeCAD stored with the document."""
  a = float(angle_deg)
  if _close(a, 360):
    return prim
  if not 0 < a <= 180 + 1e-9:
    raise UnsupportedGeometry(f'{name}: a revolution of {a} degrees (more than half a turn) is a disjunction of half-spaces: needs FreeCAD')
  kids = [prim, _half_space(0.0, size, name)]
  if not _close(a, 180):
    kids.append(_half_space(a - 180.0, size, name))
  return Node('common', children=kids, source=name)


def _primitive_of(obj):
  t = obj.TypeId
  if t == 'Part::Box':
    return Node('prim', kind=BOX, params=(obj.Length, obj.Width, obj.Height, 0.0), source=obj.Name)
  if t == 'Part::Sphere':
    R = float(obj.Radius)
    prim = Node('prim', kind=SPHERE, params=(R, 0.0, 0.0, 0.0), source=obj.Name)
    a1, a2 = float(obj.Angle1), float(obj.Angle2)
    if not (_close(a1, -90) and _close(a2, 90)):
      # a spherical segment: the sphere between the parallels at latitudes Angle1 and Angle2, closed by the planes of
      # those parallels (BRepPrim_Sphere: a meridian arc revolved, flat top and bottom)
      if not -90 - 1e-9 <= a1 < a2 <= 90 + 1e-9:
        raise UnsupportedGeometry(f'{obj.Name}: sphere latitudes {a1}, {a2}')
      z1, z2 = R * np.sin(np.radians(max(a1, -90.0))), R * np.sin(np.radians(min(a2, 90.0)))
      slab = Node('prim', placement=Placement(base=(-2 * R, -2 * R, z1)), kind=BOX, params=(4 * R, 4 * R, z2 - z1, 0.0), source=obj.Name)
      prim = Node('common', children=[prim, slab], source=obj.Name)
    swept = _swept(prim, obj.Angle3, R, obj.Name)
    if swept is not prim and prim.op == 'common':        # (one conjunction, not a nest of them)
      swept = Node('common', children=prim.children + swept.children[1:], source=obj.Name)
    return swept
  if t == 'Part::Cylinder':
    prim = Node('prim', kind=CYLINDER, params=(obj.Radius, obj.Height, 0.0, 0.0), source=obj.Name)
    return _swept(prim, obj.Angle, max(float(obj.Radius), float(obj.Height)), obj.Name)
  if t == 'Part::Cone':
    prim = Node('prim', kind=CONE, params=(obj.Radius1, obj.Radius2, obj.Height, 0.0), source=obj.Name)
    return _swept(prim, obj.Angle, max(float(obj.Radius1), float(obj.Radius2), float(obj.Height)), obj.Name)
  if t == 'Part::Torus':
    if not (_close(obj.Angle1, -180) and _close(obj.Angle2, 180)):
      # (a tube that is not closed: what OpenCASCADE closes it with -- planes from the arc's ends to the axis -- makes a
      #  solid that is not a torus any more; not derived here)
      raise UnsupportedGeometry(f'{obj.Name}: tori with an open tube section need FreeCAD')
    prim = Node('prim', kind=TORUS, params=(obj.Radius1, obj.Radius2, 0.0, 0.0), source=obj.Name)
    return _swept(prim, obj.Angle3, float(obj.Radius1) + float(obj.Radius2), obj.Name)
  if t == 'Part::FeaturePython' and obj.ProxyClass == 'Paraboloid':
    # solid paraboloid of revolution x^2 + y^2 <= 4 f z, z <= Height (freecad_elements.make.makeParaboloid;
    # FreeCAD has no such primitive: there it is the revolution of a parabola about its axis)
    f, h = float(obj.FocalLength), float(obj.Height)
    if not (f > 0 and h > 0):
      raise UnsupportedGeometry(f'{obj.Name}: paraboloid needs a positive focal length and height')
    return Node('prim', kind=PARABOLOID, params=(f, h, 2.0 * np.sqrt(f * h), 0.0), source=obj.Name, facemask=PARABOLOID_FACES)
  return None


def _moved(nodes, placement):
  out = []
  for n in nodes:
    out.append(Node(n.op, placement * n.placement, n.mesh, n.kind, n.params, n.children, n.source, n.facemask))
  return out


def _link_scale(obj):
  """the uniform scale of an App::Link (properties Scale / ScaleVector); other scales have no
  counterpart among the analytic primitives (a sphere would become an ellipsoid)"""
  v = obj._props.get('ScaleVector')
  s = obj._props.get('Scale')
  if v is not None:
    v = np.asarray(v, dtype=np.float64).ravel()
    if len(v) == 3:
      if np.abs(v - v[0]).max() > 1e-12 * max(1.0, abs(v[0])):
        raise UnsupportedGeometry(f'{obj.Name}: link with a different scale per axis {v.tolist()}')
      return float(v[0])
  return float(s) if isinstance(s, (int, float)) else 1.0


def _scaled(nodes, s, name=''):
  """trees magnified by s about the origin of their container (App::Link.Scale)"""
  if abs(s - 1.0) <= 1e-12:
    return nodes
  if not s > 0:
    raise UnsupportedGeometry(f'{name}: link scale {s}')

  def one(n):
    pl = Placement(matrix=np.vstack([np.hstack([n.placement.m[:3, :3], (n.placement.m[:3, 3] * s)[:, None]]), [0, 0, 0, 1]]))
    mesh = n.mesh
    if mesh is not None:
      mesh = (np.asarray(mesh[0]) * s,) + tuple(mesh[1:3])       # (face table and payload describe the unscaled shape)
    return Node(n.op, pl, mesh, n.kind, tuple(p * s for p in n.params), [one(c) for c in n.children], n.source, n.facemask)
  return [one(n) for n in nodes]


def _is_draft_array(obj):
  return obj.ProxyClass == 'Array' and (obj.ProxyModule or '').startswith('draftobjects')


def solids_of(obj, with_own_placement=True, _depth=0, brepFacets=False):
  """-> list of CSG trees (one per shell) of `obj`, in the coordinates of
  obj's container.  `with_own_placement=False` drops obj.Placement (an
  App::Link with LinkTransform=false replaces it by its own).
  Features the parametric recipe cannot express (partial revolutions, booleans of imported
  shapes, nested disjunctions) fall back to the shape FreeCAD computed and stored with the
  object, as long as the project is as it was saved (a property written since then may have
  changed the shape; placements are applied here and do not count)."""
  try:
    return _solids_by_recipe(obj, with_own_placement, _depth, brepFacets)
  except UnsupportedGeometry as e:
    payload = obj._props.get('Shape')
    if getattr(payload, 'data', None) and obj._doc.shapesAsSaved:
      own = obj.Placement if obj.hasProperty('Placement') and with_own_placement else Placement.identity()
      return _moved([_brep_node(obj, facets=brepFacets)], own)
    raise e


def _solids_by_recipe(obj, with_own_placement=True, _depth=0, brepFacets=False):
  if _depth > 50:
    raise UnsupportedGeometry(f'{obj.Name}: link recursion')
  own = obj.Placement if obj.hasProperty('Placement') and with_own_placement else Placement.identity()
  t = obj.TypeId

  prim = _primitive_of(obj)
  if prim is not None:
    return _moved([prim], own)
  if t == 'Mesh::Feature' and obj.hasProperty('Triangles'):
    # a tessellated shape held in memory (freecad_elements.make.makeMesh / makeTessellated, scene.stl):
    # facets counter-clockwise seen from outside, optional unit normals per vertex
    v = np.asarray(obj.Vertices, dtype=np.float64).reshape(-1, 3)
    tri = np.asarray(obj.Triangles, dtype=np.int64).reshape(-1, 3)
    vn = obj._props.get('VertexNormals')
    vn = None if vn is None else np.asarray(vn, dtype=np.float64).reshape(-1, 3)
    if len(tri) == 0 or tri.min() < 0 or tri.max() >= len(v) or (vn is not None and vn.shape != v.shape):
      raise UnsupportedGeometry(f'{obj.Name}: inconsistent mesh arrays')
    return _moved([Node('mesh', mesh=(v, tri, vn), source=obj.Name)], own)

  def one(child):
    s = solids_of(child, _depth=_depth + 1, brepFacets=brepFacets)
    if len(s) != 1:
      raise UnsupportedGeometry(f'{obj.Name}: boolean operand {child.Name} is not a single solid')
    if s[0].op == 'mesh':
      raise UnsupportedGeometry(f'{obj.Name}: booleans of tessellated shapes need FreeCAD')
    return s[0]

  if t in ('Part::MultiCommon', 'Part::MultiFuse'):
    kids = [one(c) for c in obj.Shapes]
    return _moved([Node('common' if t == 'Part::MultiCommon' else 'fuse', children=kids, source=obj.Name)], own)
  if t in ('Part::Common', 'Part::Fuse', 'Part::Cut'):
    op = {'Part::Common': 'common', 'Part::Fuse': 'fuse', 'Part::Cut': 'cut'}[t]
    return _moved([Node(op, children=[one(obj.Base), one(obj.Tool)], source=obj.Name)], own)

  if _is_draft_array(obj) or (t.startswith('App::Link') and not t.startswith('App::LinkGroup')):
    target = obj.Base if _is_draft_array(obj) else obj.LinkedObject
    if target is None:
      return []
    link_transform = bool(obj._props.get('LinkTransform', False))
    base = solids_of(target, with_own_placement=link_transform, _depth=_depth + 1, brepFacets=brepFacets)
    count = int(obj._props.get('ElementCount', 0) or 0)
    plist = obj._props.get('PlacementList') or []
    if _is_draft_array(obj):
      count = int(obj._props.get('Count', len(plist)) or len(plist))
      if not bool((obj._props.get('Proxy') or {}).get('state', {}).get('use_link', True)):
        base = solids_of(target, _depth=_depth + 1, brepFacets=brepFacets)
    base = _scaled(base, _link_scale(obj), obj.Name)
    if count > 0:
      if len(plist) < count:
        raise UnsupportedGeometry(f'{obj.Name}: array without stored PlacementList')
      scales = obj._props.get('ScaleList')
      if isinstance(scales, list) and any(np.abs(np.asarray(v, float) - 1.0).max() > 1e-12 for v in scales[:count]):
        raise UnsupportedGeometry(f'{obj.Name}: link array with scaled elements')
      out = []
      for pl in plist[:count]:
        out.extend(_moved(base, pl))
      return _moved(out, own)
    return _moved(base, own)

  if t.startswith('App::LinkGroup') or t in ('App::Part', 'App::DocumentObjectGroup', 'Part::Compound'):
    key = 'ElementList' if t.startswith('App::LinkGroup') else ('Links' if t == 'Part::Compound' else 'Group')
    out = []
    for c in obj._props.get(key) or []:
      out.extend(solids_of(c, _depth=_depth + 1, brepFacets=brepFacets))
    return _moved(out, own)

  if t in ('Part::Feature', 'Part::FeaturePython') or t.startswith('PartDesign') or t.startswith('Sketcher'):
    return _moved([_brep_node(obj, facets=brepFacets)], own)
  return []


BREP_DEFLECTION = 1e-3      # mm between a facet and the exact surface (`Shape.tessellate(tol)`)
_BREP_CACHE = {}


BREP_EXACT = True           # solids bounded by quadrics / tori only: exact CSG instead of facets


def _brep_node(obj, facets=False):
  """objects without a parametric recipe (STEP imports, PartDesign bodies): their stored
  boundary representation, in the object's own coordinates (FreeCAD keeps the Placement as the
  location of the stored shape: it is taken off here and applied by the caller) -- as exact
  CSG over the analytic primitives where the solid is an intersection of quadric half-spaces
  (scene/brep_csg.py: catalogue lenses, prisms, plates), as facets otherwise or on request"""
  from . import brep, brep_csg, brep_mesh
  payload = obj._props.get('Shape')
  if payload is None or not getattr(payload, 'data', None):
    raise UnsupportedGeometry(f'{obj.Name} ({obj.TypeId}): no parametric recipe and no stored BRep payload')
  key = (payload.name, len(payload.data), hash(payload.data), BREP_DEFLECTION)
  if key not in _BREP_CACHE:
    try:
      parsed = brep.load(payload.data)
      m = brep_mesh.tessellate(parsed, deflection=BREP_DEFLECTION, keep_root_location=False)
      exact = brep_csg.recognise(parsed, m)
    except brep.BRepError as e:
      raise UnsupportedGeometry(f'{obj.Name} ({obj.TypeId}): BRep payload {payload.name}: {e}') from e
    _BREP_CACHE[key] = (m, exact)
  m, exact = _BREP_CACHE[key]
  if exact is not None and BREP_EXACT and not facets:
    return _named(_copy_tree(exact[0]), obj.Name)
  return Node('mesh', mesh=(m.vertices, m.triangles, m.normals, m.faces, payload), source=obj.Name)


def _copy_tree(n):
  return Node(n.op, n.placement, n.mesh, n.kind, n.params, [_copy_tree(c) for c in n.children], n.source, n.facemask)


def _named(n, name):
  n.source = name
  for c in n.children:
    _named(c, name)
  return n


# ---------------------------------------------------------------------------
# flattening a CSG tree into primitives with trimming conditions
# ---------------------------------------------------------------------------
@dataclass
class FlatPrim:
  kind: int
  params: tuple
  to_world: Placement
  flip: bool
  conds: list            # [(FlatPrim, want_inside)]
  facemask: int
  source: str
  index: int = -1


def _leaves(node, acc, out):
  pl = acc * node.placement
  if node.op == 'prim':
    fp = FlatPrim(node.kind, tuple(float(p) for p in node.params), pl, False, [],
                  ((1 << N_FACES[node.kind]) - 1) & node.facemask, node.source)
    node._flat = fp
    out.append(fp)
  else:
    for c in node.children:
      _leaves(c, pl, out)


def _inside_conj(node):
  if node.op == 'prim':
    return [(node._flat, True)]
  if node.op == 'common':
    return [c for k in node.children for c in _inside_conj(k)]
  if node.op == 'cut':
    return _inside_conj(node.children[0]) + _outside_conj(node.children[1])
  raise UnsupportedGeometry(f'{node.source}: "inside a Fuse" is a disjunction; nested this way it needs FreeCAD')


def _outside_conj(node):
  if node.op == 'prim':
    return [(node._flat, False)]
  if node.op == 'fuse':
    return [c for k in node.children for c in _outside_conj(k)]
  raise UnsupportedGeometry(f'{node.source}: "outside a {node.op}" is a disjunction; nested this way it needs FreeCAD')


def _assign(node, conds, flip):
  if node.op == 'prim':
    node._flat.conds = list(conds)
    node._flat.flip = flip
    return
  kids = node.children
  if node.op == 'common':
    for i, k in enumerate(kids):
      extra = [c for j, o in enumerate(kids) if j != i for c in _inside_conj(o)]
      _assign(k, conds + extra, flip)
  elif node.op == 'fuse':
    for i, k in enumerate(kids):
      extra = [c for j, o in enumerate(kids) if j != i for c in _outside_conj(o)]
      _assign(k, conds + extra, flip)
  elif node.op == 'cut':
    _assign(kids[0], conds + _outside_conj(kids[1]), flip)
    _assign(kids[1], conds + _inside_conj(kids[0]), not flip)


def is_convex(node):
  """True for solids a straight line meets in one interval: box, sphere,
  cylinder, cone and intersections (Common) of such"""
  if node.op == 'prim':
    return node.kind in (BOX, SPHERE, CYLINDER, CONE, PARABOLOID)
  if node.op == 'common':
    return all(is_convex(c) for c in node.children)
  return False


def mesh_is_convex(vertices, triangles):
  """`mesh_convexity` as a yes / no"""
  return mesh_convexity(vertices, triangles) > 0


def mesh_convexity(vertices, triangles):
  """0: not (known to be) convex; 1: convex by the test below, whose edges may bend OUTWARD by up to 1e-9 of the mesh's
  size (what many small bends add up to across a fine mesh is not bounded by it); 2: no edge bends outward by more than
  rounding (1e-13 of the size) -- the polyhedron is convex as it stands, every point of a facet lies on or below the
  plane of every other facet up to rounding (ODW_FLAG_STRICTLY_CONVEX: the mesh kernel's normal cones rely on it).

  Is the tessellated shape the boundary of a convex solid, facets counter-clockwise seen from outside?  A closed
  surface (every edge in exactly two facets, once in each direction, after welding coincident vertices -- seams,
  poles) whose edges are all convex (the vertex opposite the edge in one facet is not above the plane of the other)
  bounds a convex body; the enclosed volume is positive for outward facets.  Facets without area (pole fans) are left
  out.  The tracer's "a ray that has left a convex solid cannot meet it again" rule then holds for it exactly
  (decided with the FACET's normal, not the interpolated one)."""
  v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
  tri = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
  if len(tri) < 4:
    return 0
  size = float(np.ptp(v, axis=0).max())
  if not size > 0:
    return 0
  q = np.round(v / (1e-9 * size)).astype(np.int64)
  o = np.lexsort((q[:, 2], q[:, 1], q[:, 0]))
  first = np.ones(len(q), dtype=bool)
  first[1:] = np.any(q[o][1:] != q[o][:-1], axis=1)
  weld = np.empty(len(q), dtype=np.int64)
  weld[o] = np.cumsum(first) - 1
  t = weld.reshape(-1)[tri]
  a, b, c = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
  nrm = np.cross(b - a, c - a)
  area2 = np.linalg.norm(nrm, axis=1)
  keep = (area2 > 1e-14 * size * size) & (t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 2] != t[:, 0])
  t, a, nrm, area2 = t[keep], a[keep], nrm[keep], area2[keep]
  if len(t) < 4:
    return 0
  # directed edges (from, to) with their facet and the vertex opposite
  e_from = np.concatenate([t[:, 0], t[:, 1], t[:, 2]])
  e_to = np.concatenate([t[:, 1], t[:, 2], t[:, 0]])
  e_opp = np.concatenate([t[:, 2], t[:, 0], t[:, 1]])
  e_face = np.tile(np.arange(len(t)), 3)
  n_v = int(weld.max()) + 1
  key = e_from * n_v + e_to
  rev = e_to * n_v + e_from
  order = np.argsort(key)
  ks = key[order]
  if np.any(ks[1:] == ks[:-1]):
    return 0                                       # an edge used twice in the same direction
  pos = np.searchsorted(ks, rev)
  if np.any(pos >= len(ks)) or np.any(ks[np.minimum(pos, len(ks) - 1)] != rev):
    return 0                                       # open: an edge without its opposite
  mate = order[pos]                                # the same edge in the neighbouring facet
  wpos = np.zeros((n_v, 3))
  wpos[weld.reshape(-1)] = v                       # (one representative per welded vertex)
  unit = nrm / area2[:, None]
  worst = -np.inf
  for k in range(3):                               # (edge k of every facet: a third of the gathers at a time)
    sl = slice(k * len(t), (k + 1) * len(t))
    above = np.einsum('ij,ij->i', unit, wpos[e_opp[mate[sl]]] - a)
    worst = max(worst, float(above.max()))
    if worst > 1e-9 * size:
      return 0
  volume = np.einsum('ij,ij->i', a, nrm).sum() / 6.0
  if not volume > 0:
    return 0
  # (facets without area were left out above: a mesh that has any keeps the plain answer)
  return 2 if (worst <= 1e-13 * size and bool(keep.all())) else 1


_CONVEX_MEMO = []          # [(vertices, triangles, answer)]: the arrays of stored shapes are cached objects, bakes repeat


def meshConvex(mesh):
  """mesh_is_convex of a 'mesh' node's arrays (local coordinates: a placement does not change the answer), remembered
  for the array objects seen last (a parameter sweep bakes the same stored shape again and again)"""
  v, tri = mesh[0], mesh[1]
  for mv, mt, ans in _CONVEX_MEMO:
    if mv is v and mt is tri:
      return ans
  ans = mesh_convexity(v, tri)                     # (0 / 1 / 2: truth value = mesh_is_convex)
  _CONVEX_MEMO.append((v, tri, ans))
  del _CONVEX_MEMO[:-8]
  return ans


def flatten(tree, acc=None):
  """CSG tree -> [FlatPrim] (every leaf once; conditions reference leaves)"""
  out = []
  _leaves(tree, acc or Placement.identity(), out)
  _assign(tree, [], False)
  _prune_faces(out)
  return out


# ---------------------------------------------------------------------------
# bounding boxes (world AABB) of primitives and of their faces
# ---------------------------------------------------------------------------
def local_bounds(kind, params):
  p = params
  if kind == BOX:
    return np.array([0, 0, 0.0]), np.array([p[0], p[1], p[2]])
  if kind == SPHERE:
    return np.full(3, -p[0]), np.full(3, p[0])
  if kind == CYLINDER:
    return np.array([-p[0], -p[0], 0.0]), np.array([p[0], p[0], p[1]])
  if kind == CONE:
    r = max(p[0], p[1])
    return np.array([-r, -r, 0.0]), np.array([r, r, p[2]])
  if kind == TORUS:
    r = p[0] + p[1]
    return np.array([-r, -r, -p[1]]), np.array([r, r, p[1]])
  if kind == PARABOLOID:
    r = 2.0 * np.sqrt(p[0] * p[1])
    return np.array([-r, -r, 0.0]), np.array([r, r, p[1]])
  raise ValueError(kind)


def face_local_bounds(kind, params, face):
  lo, hi = local_bounds(kind, params)
  lo, hi = lo.copy(), hi.copy()
  if kind == BOX:
    a = face >> 1
    v = hi[a] if face & 1 else lo[a]
    lo[a] = hi[a] = v
  elif kind == PARABOLOID and face == 2:
    r = 2.0 * np.sqrt(params[0] * params[1])
    lo = np.array([-r, -r, hi[2]]); hi = np.array([r, r, hi[2]])
  elif kind in (CYLINDER, CONE) and face in (1, 2):
    r = params[0] if (face == 1 or kind == CYLINDER) else params[1]
    z = lo[2] if face == 1 else hi[2]
    lo = np.array([-r, -r, z]); hi = np.array([r, r, z])
  return lo, hi


_CORNER_IS_HI = np.array([[x, y, z] for x in (False, True) for y in (False, True) for z in (False, True)])


def world_aabb(to_world, lo, hi):
  corners = np.where(_CORNER_IS_HI, hi, lo)                # the eight corners, x slowest
  w = corners @ to_world.m[:3, :3].T + to_world.m[:3, 3]
  return w.min(axis=0), w.max(axis=0)


def _prune_faces(prims, slack=1e-3):
  """drop faces that can never satisfy a must-be-inside condition (their
  bounding box misses the other operand's): an optimisation only, the trim
  test would reject every candidate on them anyway"""
  boxes = {}                                               # world boxes of the operands, each worked out once

  def box_of(p):
    b = boxes.get(id(p))
    if b is None:
      b = boxes[id(p)] = world_aabb(p.to_world, *local_bounds(p.kind, p.params))
    return b

  for fp in prims:
    must_be_in = [other for other, inside in fp.conds if inside]
    if not must_be_in:                                     # (nothing to miss: every face it has stays)
      fp.facemask &= (1 << N_FACES[fp.kind]) - 1
      continue
    mask = 0
    for f in range(N_FACES[fp.kind]):
      if not (fp.facemask >> f) & 1:
        continue
      flo, fhi = world_aabb(fp.to_world, *face_local_bounds(fp.kind, fp.params, f))
      keep = True
      for other in must_be_in:
        olo, ohi = box_of(other)
        if np.any(flo > ohi + slack) or np.any(fhi < olo - slack):
          keep = False
          break
      if keep:
        mask |= 1 << f
    fp.facemask = mask


# ---------------------------------------------------------------------------
# tessellation of the analytic primitives (what FreeCAD's Shape.tessellate
# returns for such faces): test vehicle of the triangle path and the form in
# which shapes with non-quadric surfaces reach the tracer
# ---------------------------------------------------------------------------
def meshWorld(node):
  """world-space arrays of a 'mesh' node: vertices (n,3), triangles, normals"""
  v, tri, vn = node.mesh[:3]     # (BRep shapes carry their face table and payload as further entries)
  R, t = node.placement.m[:3, :3], node.placement.m[:3, 3]
  return v @ R.T + t, tri, (None if vn is None else vn @ R.T)


def _grid(nu, nv, point, normal, wrap_u=True):
  """triangulated (u, v) grid: point(u, v), normal(u, v) with u in [0,1) periodic"""
  us = np.linspace(0, 1, nu + 1)
  vs = np.linspace(0, 1, nv + 1)
  U, V = np.meshgrid(us, vs, indexing='ij')
  if wrap_u:
    U[-1, :] = U[0, :]            # periodic in u: the seam vertices coincide exactly (watertight)
  P, N = point(U, V).reshape(-1, 3), normal(U, V).reshape(-1, 3)
  idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
  a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
  tri = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
  return P, tri, N


def tessellate(kind, params, segments=48):
  """-> (vertices, triangles, vertex normals) of a primitive in its local frame;
  facets counter-clockwise seen from outside; every face has its own vertices
  (no normal smoothing across edges).  Degenerate facets (poles, apex) are dropped."""
  n = int(segments)
  two_pi = 2 * np.pi
  parts = []
  st = lambda *a: np.stack(np.broadcast_arrays(*a), axis=-1)
  if kind == SPHERE:
    R = params[0]
    m = max(2, n // 2)
    pt = lambda U, V: st(R * np.sin(np.pi * V) * np.cos(two_pi * U), R * np.sin(np.pi * V) * np.sin(two_pi * U),
                         -R * np.cos(np.pi * V))
    parts.append(_grid(n, m, pt, lambda U, V: pt(U, V) / R))
  elif kind == TORUS:
    R1, R2 = params[0], params[1]
    nr = lambda U, V: st(np.cos(two_pi * V) * np.cos(two_pi * U), np.cos(two_pi * V) * np.sin(two_pi * U),
                         np.sin(two_pi * V))
    pt = lambda U, V: st((R1 + R2 * np.cos(two_pi * V)) * np.cos(two_pi * U),
                         (R1 + R2 * np.cos(two_pi * V)) * np.sin(two_pi * U), R2 * np.sin(two_pi * V))
    parts.append(_grid(n, max(3, n // 2), pt, nr))
  elif kind in (CYLINDER, CONE):
    r1, r2, h = (params[0], params[0], params[1]) if kind == CYLINDER else (params[0], params[1], params[2])
    k = (r2 - r1) / h
    inv = 1 / np.sqrt(1 + k * k)
    pt = lambda U, V: st((r1 + k * h * V) * np.cos(two_pi * U), (r1 + k * h * V) * np.sin(two_pi * U), h * V)
    nr = lambda U, V: st(np.cos(two_pi * U) * inv, np.sin(two_pi * U) * inv, -k * inv + 0 * V)
    parts.append(_grid(n, max(1, n // 8), pt, nr))
    for r, z, s in ((r1, 0.0, -1.0), (r2, h, 1.0)):
      if r > 0:
        # disc: u angle, v radius; (d/du) x (d/dv) = tangent x radial: clockwise u gives +z
        dp = lambda U, V, r=r, z=z, s=s: st(r * V * np.cos(-s * two_pi * U), r * V * np.sin(-s * two_pi * U), z + 0 * V)
        dn = lambda U, V, s=s: st(0 * U, 0 * U, s + 0 * V)
        parts.append(_grid(n, max(1, n // 8), dp, dn))
  elif kind == PARABOLOID:
    f, h = params[0], params[1]
    rim = 2.0 * np.sqrt(f * h)
    # v = radius / rim (equal steps in radius: the chord error is even along the meridian)
    pt = lambda U, V: st(rim * V * np.cos(two_pi * U), rim * V * np.sin(two_pi * U), (rim * V)**2 / (4 * f))
    def nr(U, V):
      g = st(rim * V * np.cos(two_pi * U), rim * V * np.sin(two_pi * U), -2.0 * f + 0 * V)
      return g / np.linalg.norm(g, axis=-1, keepdims=True)
    # (d/du) x (d/dv) must point outwards (away from the axis, downwards): u clockwise
    lat = _grid(n, max(2, n // 4), lambda U, V: pt(-U, V), lambda U, V: nr(-U, V))
    parts.append(lat)
    dp = lambda U, V: st(rim * V * np.cos(-two_pi * U), rim * V * np.sin(-two_pi * U), h + 0 * V)
    parts.append(_grid(n, max(1, n // 8), dp, lambda U, V: st(0 * U, 0 * U, 1.0 + 0 * V)))
  elif kind == BOX:
    L = np.array(params[:3], dtype=np.float64)
    for a in range(3):
      b1, b2 = (a + 1) % 3, (a + 2) % 3
      for s in (0, 1):
        def pt(U, V, a=a, b1=b1, b2=b2, s=s):
          c = [None] * 3
          c[a] = L[a] * s + 0 * U
          # (b1, b2, a) is right-handed: u along b1, v along b2 gives normal +a; swap for the low face
          c[b1], c[b2] = (L[b1] * U, L[b2] * V) if s else (L[b1] * V, L[b2] * U)
          return st(*c)
        def nr(U, V, a=a, s=s):
          c = [0 * U, 0 * U, 0 * U]
          c[a] = (1.0 if s else -1.0) + 0 * U
          return st(*c)
        parts.append(_grid(1, 1, pt, nr, wrap_u=False))
  else:
    raise UnsupportedGeometry(f'cannot tessellate primitive kind {kind}')
  V, T, N, off = [], [], [], 0
  for p, t, nn in parts:
    V.append(p); T.append(t + off); N.append(nn)
    off += len(p)
  V, T, N = np.concatenate(V), np.concatenate(T), np.concatenate(N)
  e1, e2 = V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]]
  area2 = np.linalg.norm(np.cross(e1, e2), axis=1)
  return V, T[area2 > 1e-12 * max(1.0, float(np.abs(V).max()))**2], N
