"""One-time scene bake: document -> flat SoA tables for the device.

This is the build's counterpart of the reference's per-run caches
(simulation/raytracing_cache.py:43-114) and per-segment document scans
(freecad_elements/find.py:59-141): everything static is resolved once --
global placements (freecad_elements/common.py:36-125), the optical property
table (freecad_elements/optical_group.py:29-96), the tracing sequence
(freecad_elements/simulation_settings.py:158-196) and the limits of
Ray.traceRay (freecad_elements/ray.py:46-73, 283-288).
"""
from dataclasses import dataclass, field

import numpy as np

from . import geometry
from .placement import Placement

OPTICAL_TYPES = ['Mirror', 'Lens', 'Grating', 'Absorber', 'Vacuum']
MAX_GROUPS = 64
_SOURCE_PROXIES = ('PointSourceProxy', 'SurfaceSourceProxy', 'ReplaySourceProxy')
_UNSUPPORTED_SOURCES = ()


# ---------------------------------------------------------------------------
# find.* equivalents (freecad_elements/find.py)
# ---------------------------------------------------------------------------
def lightSources(doc):
  # (all find.* functions look through the simulating document and every document its links lead
  #  into: find._allObjects, freecad_elements/find.py:24-56)
  return [o for o in doc.allObjects()
          if o.TypeId == 'App::LinkGroupPython' and o.ProxyClass in _SOURCE_PROXIES + _UNSUPPORTED_SOURCES]


def opticalObjects(doc):
  return [o for o in doc.allObjects()
          if o.TypeId == 'App::LinkGroupPython' and o.ProxyClass == 'OpticalGroupProxy']


def simulationSettings(doc):
  return [o for o in doc.allObjects()
          if o.TypeId == 'Part::FeaturePython' and o.ProxyClass == 'SimulationSettingsProxy']


def activeSimulationSettings(doc):
  """find.py:116-141: the single Active settings object, else the first one"""
  allset = simulationSettings(doc)
  active = [s for s in allset if s._props.get('Active', True)]
  if len(active) > 1:
    raise ValueError('only one simulation settings object may have its "Active" property '
                     'set to true, but the following objects are active: '
                     + ', '.join(o.Name for o in active))
  if active:
    return active[0]
  return allset[0] if allset else None


def _is_link(o):
  return o.TypeId.startswith('App::Link') and not o.TypeId.startswith('App::LinkGroup')


def allPlacementsAndPaths(doc, obj, ignoreLinks=False, _depth=0):
  """every representation of `obj` in the global model: [(Placement, path)]
  with path = tuple of object names from the top level down to obj.
  Follows allPlacementsAndPaths (freecad_elements/common.py:36-109):
  containers (App::Part / LinkGroup / DocumentObjectGroup) compose their
  placement with the child's; an App::Link pointing at obj contributes the
  link's own placements, replacing obj.Placement unless LinkTransform is set.
  DocumentObjectGroups have no placement and are transparent: an object listed
  in a group and in the Part around that group is one instance (paths are
  compared without groups).
  Objects of documents the project links into are represented through those
  links only: a chain that ends at the top level of another document is not part
  of the global model (common.py:62-65)."""
  if _depth > 100:
    raise RuntimeError('allPlacementsAndPaths reached recursion depth 100')
  own = obj.Placement if obj.hasProperty('Placement') else Placement.identity()
  out = []
  containers = obj._doc.parents_of(obj)
  links = [] if ignoreLinks else [o for o in doc.allObjects()
                                  if _is_link(o) and o._props.get('LinkedObject') is obj]
  if not containers and obj._doc is doc:
    out.append((own, (obj.Name,)))
  for c in containers:
    transparent = c.TypeId == 'App::DocumentObjectGroup'
    for pl, path in allPlacementsAndPaths(doc, c, ignoreLinks, _depth + 1):
      out.append((pl * own, (path[:-1] if transparent else path) + (obj.Name,)))
  for l in links:
    if abs(geometry._link_scale(l) - 1.0) > 1e-12:
      # (solids inside a scaled link are scaled by geometry.solids_of; an optical group or a light
      #  source there would need a placement that is not rigid)
      raise geometry.UnsupportedGeometry(f'{obj.Name} is reached through the scaled link {l.Name}')
    keep_own = own if bool(l._props.get('LinkTransform', False)) else Placement.identity()
    for pl, path in allPlacementsAndPaths(doc, l, ignoreLinks, _depth + 1):
      out.append((pl * keep_own, path + (obj.Name,)))
  seen, uniq = set(), []
  for pl, path in out:
    if path not in seen:
      seen.add(path)
      uniq.append((pl, path))
  return sorted(uniq, key=lambda e: '.'.join(e[1])) if _depth == 0 else uniq


def globalPlacements(doc, obj, ignoreLinks=False):
  """global placements of `obj` (one per representation)"""
  return [pl for pl, _ in allPlacementsAndPaths(doc, obj, ignoreLinks)]


def allCoordinateTransformMatrices(doc, obj, ignoreLinks=False):
  """[gpM, gpMi, pM, pMi] (4x4 arrays) per representation of obj
  (freecad_elements/common.py:110-123)"""
  own = obj.Placement if obj.hasProperty('Placement') else Placement.identity()
  return [[pl.m.copy(), pl.inverse().m.copy(), own.m.copy(), own.inverse().m.copy()]
          for pl, _ in allPlacementsAndPaths(doc, obj, ignoreLinks)]


_GLOBAL_INFO_SKIP = set('Placement Proxy Shape ShapeMaterial ColoredElements ElementList ExpressionEngine '
                        'LinkedChildren VisibilityList Visibility Height Length Width MapMode MapPathParameter '
                        'MapReversed'.split())


def _exportable(value):
  """property values as the reference exports them (freecad_elements/__init__.py:55-91):
  document objects by their exported properties, lists element-wise, other values as they are"""
  from .fcstd import DocumentObject
  if isinstance(value, DocumentObject):
    return _propertiesToDict(value)
  if isinstance(value, (list, tuple)):
    return [_exportable(v) for v in value]
  if isinstance(value, np.ndarray):
    return value.copy()
  if isinstance(value, Placement):
    return str(value)
  return value


def _propertiesToDict(obj):
  if obj is None:
    return None
  res = {k: _exportable(v) for k, v in obj._props.items()
         if not k.startswith('_') and not k.startswith('Attach') and k not in _GLOBAL_INFO_SKIP}
  res['Name'] = obj.Name
  return res


def collectGlobalInfo(doc):
  """globally relevant info about the simulation project, the content of
  `global-info.pkl` (freecad_elements/__init__.py:48-115): active settings,
  light sources (placements without links) and optical objects with their
  properties, placement paths and transformation matrices"""

  def objToDict(obj, ignoreLinks=False):
    props = _propertiesToDict(obj)
    reps = allPlacementsAndPaths(doc, obj, ignoreLinks)
    mats = allCoordinateTransformMatrices(doc, obj, ignoreLinks)
    return dict(name=props.pop('Name'), label=props.pop('Label', obj.Name), properties=props,
                placementPathsAndMatrices=[dict(path=list(path), gpM=m[0], gpMi=m[1], pM=m[2], pMi=m[3])
                                           for (_, path), m in zip(reps, mats)])

  return dict(activeSimulationSettings=_propertiesToDict(activeSimulationSettings(doc)),
              lightSources=[objToDict(s, ignoreLinks=True) for s in lightSources(doc)],
              opticalObjects=[objToDict(o) for o in opticalObjects(doc)])


def tracingSequence(settings):
  """SimulationSettingsProxy.getTracingSequence (simulation_settings.py:158-196):
  non-empty SequentialModeElements_NN lists in ascending order"""
  if settings is None or not settings._props.get('SequentialMode', False):
    return []
  seq = []
  for i in range(100):
    lst = settings._props.get(f'SequentialModeElements_{i:02d}')
    if lst:
      seq.append(list(lst))
  return seq


# ---------------------------------------------------------------------------
@dataclass
class BakedScene:
  prim_type: np.ndarray
  prim_group: np.ndarray
  prim_solid: np.ndarray
  prim_flags: np.ndarray
  prim_xform: np.ndarray      # (n,12) global -> local
  prim_params: np.ndarray     # (n,4)
  prim_cond_off: np.ndarray
  cond_prim: np.ndarray
  cond_inside: np.ndarray
  group_type: np.ndarray
  group_ior: np.ndarray
  group_refl: np.ndarray
  group_abslen: np.ndarray
  group_record: np.ndarray
  group_grating_type: np.ndarray
  group_grating_lpm: np.ndarray
  group_grating_dir: np.ndarray
  group_grating_order: np.ndarray
  seq_enabled: int
  seq_mask: np.ndarray
  ignore_mask: int
  group_names: list = field(default_factory=list)
  group_labels: list = field(default_factory=list)
  prim_sources: list = field(default_factory=list)
  prim_to_world: list = field(default_factory=list)
  surface_samplers: list = field(default_factory=list)   # freecad_elements.optical_group.BakedSurfaceSampler
  tri_normals: np.ndarray = None                         # (n_prims, 9) vertex normals of TRIANGLE primitives, or None
  tri_edges: np.ndarray = None                           # (n_prims,) bits: which facet edges are edges of the face

  @property
  def n_prims(self):
    return int(self.prim_type.shape[0])

  @property
  def n_groups(self):
    return int(self.group_type.shape[0])

  @property
  def n_faces(self):
    return int(sum(bin(int(f) >> 8).count('1') for f in self.prim_flags))

  def group_index(self, name_or_label):
    for i, (n, l) in enumerate(zip(self.group_names, self.group_labels)):
      if name_or_label in (n, l):
        return i
    raise KeyError(name_or_label)


@dataclass
class Limits:
  max_ray_length: float = 1000.0
  max_intersections: int = 100
  dist_tol: float = 1e-2
  power_tol: float = 1e-6


def _abslen(v):
  try:
    return float(v)
  except (TypeError, ValueError):
    return float('inf')


def faceEdgeBits(tri, vertices=None):
  """per facet, which of its edges belong to one facet only (bit 0: v0-v2, bit 1: v0-v1, bit 2: v1-v2):
  the edges of the tessellated face.  A face = the facets connected through shared vertex indices
  (different faces of a solid have their own vertices, so the seam between two faces is an edge of
  both); inside a face, vertices at the same place count as one (the seam of a periodic surface and
  the pole of a fan are stored once per column)."""
  tri = np.asarray(tri, dtype=np.int64)
  ids = tri
  if vertices is not None and len(tri):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    v = np.asarray(vertices, dtype=np.float64)
    n = len(v)
    rows = np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2]])
    cols = np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0]])
    _, comp = connected_components(coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)), directed=False)
    scale = max(float(np.abs(v).max()), 1e-300)
    key = np.concatenate([comp[:, None], np.round(v / (1e-9 * scale)).astype(np.int64)], axis=1)
    _, welded = np.unique(key, axis=0, return_inverse=True)
    ids = welded.ravel()[tri]
  pairs = np.stack([ids[:, [0, 2]], ids[:, [0, 1]], ids[:, [1, 2]]], axis=1)          # (m, 3, 2)
  key = np.sort(pairs, axis=2).reshape(-1, 2)
  _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
  single = (cnt[inv.ravel()] == 1).reshape(-1, 3)
  return (single[:, 0] * 1 + single[:, 1] * 2 + single[:, 2] * 4).astype(np.int32)


def bakeScene(doc, source=None, surfaceFamily=None):
  """flat tables for all optical groups of `doc` as seen by `source`"""
  groups = opticalObjects(doc)
  if len(groups) > MAX_GROUPS:
    raise geometry.UnsupportedGeometry(f'{len(groups)} optical groups (limit {MAX_GROUPS})')
  # stochastic surfaces (optical_group.py:212-323): tables per (group, kind)
  from ..freecad_elements import optical_group as _og
  surface_samplers = []
  # (groups a ray can be inside of: their refractive indices decide which n1 / n2 a lens hit can have)
  media = {gi: float(g._props.get('RefractiveIndex', 1.0)) for gi, g in enumerate(groups)
           if g._props.get('OpticalType', 'Mirror') == 'Lens'
           or (g._props.get('OpticalType') == 'Grating' and str(g._props.get('GratingType', '')).lower().startswith('trans'))}
  for gi, g in enumerate(groups):
    surface_samplers += _og.surfaceSamplers(g, gi, n_family=surfaceFamily or _og.DEFAULT_FAMILY, media=media)
  prims = []
  prim_group, prim_solid = [], []
  meshes = []                    # (group, solid id, world vertices, triangles, world normals or None, source)
  solid_id = 0
  for gi, g in enumerate(groups):
    for gp in globalPlacements(doc, g):
      # the group's children are placed relative to the group; gp already
      # contains the group's own placement
      for child in g._props.get('ElementList') or []:
        for tree in geometry.solids_of(child):
          if tree.op == 'mesh':
            node = geometry.Node('mesh', gp * tree.placement, tree.mesh, source=tree.source)
            meshes.append((gi, solid_id) + geometry.meshWorld(node) + (tree.source, geometry.meshConvex(tree.mesh)))
            solid_id += 1
            continue
          flat = geometry.flatten(tree, gp)
          convex = geometry.is_convex(tree)
          for fp in flat:
            fp.convex = convex
            fp.index = len(prims)
            prims.append(fp)
            prim_group.append(gi)
            prim_solid.append(solid_id)
          solid_id += 1
  n = len(prims)
  cond_off, cond_prim, cond_inside = [0], [], []
  for fp in prims:
    for other, inside in fp.conds:
      cond_prim.append(other.index)
      cond_inside.append(1 if inside else 0)
    cond_off.append(len(cond_prim))

  def col(key, default, dtype=np.float64):
    return np.array([g._props.get(key, default) for g in groups], dtype=dtype)

  types = []
  for g in groups:
    t = g._props.get('OpticalType', 'Mirror')
    types.append(OPTICAL_TYPES.index(t) if isinstance(t, str) else int(t))
  gt = []
  for g in groups:
    t = g._props.get('GratingType', 'Reflection')
    gt.append(['Reflection', 'Transmission'].index(t) if isinstance(t, str) else int(t))
  gdir = np.array([np.asarray(g._props.get('GratingLinesOrientation', (0, 0, 1)), dtype=np.float64)
                   for g in groups], dtype=np.float64).reshape(len(groups), 3)

  settings = activeSimulationSettings(doc)
  seq = tracingSequence(settings)
  seq_mask = []
  for step in seq:
    m = 0
    for o in step:
      if o in groups:
        m |= 1 << groups.index(o)
    seq_mask.append(m)
  ignore = 0
  if source is not None:
    for o in source._props.get('IgnoredOpticalElements') or []:
      if o in groups:
        ignore |= 1 << groups.index(o)

  # tessellated shapes: one TRIANGLE primitive per facet, appended after the analytic ones
  # (vertices in global coordinates in the 12 doubles of the frame; no conditions)
  n_tri = sum(len(m[3]) for m in meshes)
  tri_xform = np.zeros((n_tri, 12))
  tri_group, tri_solid = np.zeros(n_tri, dtype=np.int32), np.zeros(n_tri, dtype=np.int32)
  tri_flags = np.full(n_tri, 1 << 8, dtype=np.int32)
  tri_normals = np.zeros((n + n_tri, 9)) if any(m[4] is not None for m in meshes) else None
  tri_edges = np.full(n + n_tri, 7, dtype=np.int32) if n_tri else None
  at = 0
  for gi, sid, v, tri, vn, _, convex in meshes:
    k = len(tri)
    tri_edges[n + at:n + at + k] = faceEdgeBits(tri, v)
    tri_xform[at:at + k, 0:9] = v[tri].reshape(k, 9)
    tri_group[at:at + k], tri_solid[at:at + k] = gi, sid
    if convex:                                     # (ODW_FLAG_CONVEX: as for convex analytic solids)
      tri_flags[at:at + k] |= 2
    if convex == 2:                                # (ODW_FLAG_STRICTLY_CONVEX: geometry.mesh_convexity)
      tri_flags[at:at + k] |= 8
    if tri_normals is not None:
      if vn is not None:
        tri_normals[n + at:n + at + k] = vn[tri].reshape(k, 9)
      else:                                        # facet normals for meshes without vertex normals
        e1, e2 = v[tri[:, 1]] - v[tri[:, 0]], v[tri[:, 2]] - v[tri[:, 0]]
        fn = np.cross(e1, e2)
        tri_normals[n + at:n + at + k] = np.tile(fn / np.linalg.norm(fn, axis=1)[:, None], (1, 3))
    at += k
  cond_off = cond_off + [cond_off[-1]] * n_tri

  return BakedScene(
      tri_normals=tri_normals, tri_edges=tri_edges,
      prim_type=np.concatenate([np.array([p.kind for p in prims], dtype=np.int32),
                                np.full(n_tri, geometry.TRIANGLE, dtype=np.int32)]),
      prim_group=np.concatenate([np.array(prim_group, dtype=np.int32), tri_group]),
      prim_solid=np.concatenate([np.array(prim_solid, dtype=np.int32), tri_solid]),
      prim_flags=np.concatenate([np.array([(1 if p.flip else 0) | (2 if getattr(p, 'convex', False) else 0) | (p.facemask << 8)
                                           for p in prims], dtype=np.int32),
                                 tri_flags]),
      prim_xform=np.concatenate([_snapFrames(np.array([p.to_world.inverse().rows12() for p in prims],
                                                      dtype=np.float64).reshape(n, 12)), tri_xform]),
      prim_params=np.concatenate([np.array([p.params for p in prims], dtype=np.float64).reshape(n, 4),
                                  np.zeros((n_tri, 4))]),
      prim_cond_off=np.array(cond_off, dtype=np.int32),
      cond_prim=np.array(cond_prim, dtype=np.int32),
      cond_inside=np.array(cond_inside, dtype=np.int32),
      group_type=np.array(types, dtype=np.int32),
      group_ior=col('RefractiveIndex', 2.0),
      group_refl=col('Reflectivity', 1.0),
      group_abslen=np.array([_abslen(g._props.get('AbsorptionLength', 'inf')) for g in groups]),
      group_record=np.array([1 if g._props.get('RecordHits', False) else 0 for g in groups], dtype=np.int32),
      group_grating_type=np.array(gt, dtype=np.int32),
      group_grating_lpm=col('GratingLinesPerMillimeter', 1000.0),
      group_grating_dir=gdir,
      group_grating_order=col('GratingDiffractionOrder', 1, dtype=np.int32),
      seq_enabled=1 if (settings is not None and settings._props.get('SequentialMode', False)) else 0,
      seq_mask=np.array(seq_mask, dtype=np.uint64),
      ignore_mask=ignore,
      group_names=[g.Name for g in groups],
      group_labels=[g._props.get('Label', g.Name) for g in groups],
      prim_sources=[p.source for p in prims] + [m[5] for m in meshes for _ in range(len(m[3]))],
      prim_to_world=[p.to_world for p in prims] + [None] * n_tri,
      surface_samplers=surface_samplers,
  )


def _snapFrames(rows):
  """Rotation entries of the global -> local frames that differ from 0 or +-1 by rounding of the
  placement chain only (|.| < 1e-15: cos 90 deg = 6e-17, products of such) are set to the exact value.
  The frame stays orthonormal to the last bit or two; an exact 0 / +-1 is a term the scene-compiled
  kernels leave out (odw_kernels.hip: xf_comb) and costs the generic ones nothing."""
  rows = np.array(rows, dtype=np.float64).reshape(-1, 12)
  rot = [0, 1, 2, 4, 5, 6, 8, 9, 10]
  r = rows[:, rot]
  r[np.abs(r) < 1e-15] = 0.0
  r[np.abs(r - 1.0) < 1e-15] = 1.0
  r[np.abs(r + 1.0) < 1e-15] = -1.0
  rows[:, rot] = r
  return rows


def bakeLimits(doc, source=None, maxRayLength=None, maxIntersections=None, powerTol=1e-6,
               distTol=None):
  """Ray.traceRay argument resolution (ray.py:46-73) and _getDistTol
  (ray.py:283-288)"""
  settings = activeSimulationSettings(doc)
  ray_scale = float(source._props.get('MaxRayLengthScale', 1)) if source is not None else 1.0
  int_scale = float(source._props.get('MaxIntersectionsScale', 1)) if source is not None else 1.0
  if settings is not None:
    if maxRayLength is None:
      maxRayLength = ray_scale * float(settings._props.get('MaxRayLength', 1000))
    if maxIntersections is None:
      maxIntersections = int_scale * float(settings._props.get('MaxIntersections', 100))
  else:
    if maxRayLength is None:
      maxRayLength = 1000 * ray_scale
    if maxIntersections is None:
      maxIntersections = 100 * int_scale
  if distTol is None:
    distTol = 1e-2
    if settings is not None:
      distTol = float(settings._props.get('DistanceTolerance', '1e-6'))
  # `numIntersections >= maxIntersections` with a float limit: the loop runs
  # ceil(maxIntersections) times
  return Limits(max_ray_length=float(maxRayLength),
                max_intersections=int(np.ceil(maxIntersections)),
                dist_tol=max(float(distTol), 1e-6), power_tol=float(powerTol))
