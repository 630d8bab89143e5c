"""Surface light source: faces of solids emit rays.

Host-side mirror of `SurfaceSourceProxy` (freecad_elements/surface_source.py):
  _generateRays   :414-553  faces of `ActiveSurfaces` (whole bodies or selected
                            faces, every global placement of the body), face
                            chosen by area, uniformly random point, polar
                            angle from the scalar random variable of
                            `PowerDensity` over `ThetaDomain`, azimuth uniform
  _makeRay        :87-108   direction = Rot(normal, phi) Rot(tangent, theta) normal
The Monte-Carlo branch runs on the device (csrc/odw_kernels.hip
`odw_emit_kernel`); this module bakes its inputs.  The reference samples (u, v)
from a refined grid of area elements of the OCC face (:269-392) and rejects
points off the trimmed face (:394-412); the analytic faces used here are
sampled exactly, with the same rejection against the boolean trimming.

Sub-element names follow OpenCASCADE's primitive builders:
  Part::Box       Face1..6 = -x, +x, -y, +y, -z, +z
  Part::Cylinder  Face1 = lateral, Face2 = top (z = H), Face3 = bottom (z = 0)
  Part::Cone      like the cylinder (faces of zero radius do not exist)
  Part::Sphere / Part::Torus   Face1
Faces of boolean results carry OCC's own numbering and cannot be selected by
name without FreeCAD; selecting such a body as a whole works.
"""
from dataclasses import dataclass

import numpy as np

from .. import distributions
from ..scene import bake as _bake
from ..scene import geometry
from . import point_source


@dataclass
class BakedSurfaceSource:
  wavelength: float
  power: float
  dist_tol: float
  prim_type: np.ndarray
  prim_flags: np.ndarray
  prim_xform: np.ndarray       # (n,12) global -> local
  prim_params: np.ndarray      # (n,4)
  prim_cond_off: np.ndarray
  cond_prim: np.ndarray
  cond_inside: np.ndarray
  face_prim: np.ndarray
  face_id: np.ndarray
  face_area: np.ndarray        # untrimmed
  t_edges: np.ndarray
  t_cdf: np.ndarray
  name: str = ''
  label: str = ''
  rays_per_iteration_scale: float = 1.0
  prim_to_world: list = None
  tri_normals: np.ndarray = None   # (n,9) vertex normals of TRIANGLE rows (facets of tessellated faces)


def faceArea(kind, params, face):
  p = params
  if kind == geometry.PARABOLOID:
    raise geometry.UnsupportedGeometry('faces of a paraboloid as a surface source are not built')
  if kind == geometry.BOX:
    a = face >> 1
    return p[(a + 1) % 3] * p[(a + 2) % 3]
  if kind == geometry.SPHERE:
    return 4 * np.pi * p[0]**2
  if kind == geometry.TORUS:
    return 4 * np.pi**2 * p[0] * p[1]
  r1, r2, h = (p[0], p[0], p[1]) if kind == geometry.CYLINDER else (p[0], p[1], p[2])
  if face == 0:
    return np.pi * (r1 + r2) * np.hypot(h, r2 - r1)
  return np.pi * (r1 if face == 1 else r2)**2


def _faceIndex(kind, params, name):
  """OCC sub-element name -> face bit position"""
  if not name.startswith('Face'):
    raise geometry.UnsupportedGeometry(f'sub-element {name!r} is not a face')
  k = int(name[4:]) - 1
  if kind == geometry.BOX:
    order = [0, 1, 2, 3, 4, 5]
  elif kind in (geometry.SPHERE, geometry.TORUS):
    order = [0]
  else:
    r1, r2 = (params[0], params[0]) if kind == geometry.CYLINDER else (params[0], params[1])
    order = [0] + ([2] if r2 > 0 else []) + ([1] if r1 > 0 else [])
  if not 0 <= k < len(order):
    raise geometry.UnsupportedGeometry(f'{name}: the solid has {len(order)} faces')
  return order[k]


def distTol(doc):
  """SurfaceSourceProxy._getDistTol (surface_source.py:111-116)"""
  settings = _bake.activeSimulationSettings(doc)
  tol = float(settings._props.get('DistanceTolerance', '1e-6')) if settings is not None else 1e-6
  return max(tol, 1e-9)


def _meshFacets(obj, part, tree, container, subs):
  """facets of the selected faces of a tessellated shape (BRep import), in global coordinates:
  -> (corners (k, 9), vertex normals (k, 9) or None, areas (k,))"""
  v, tri, vn = geometry.meshWorld(geometry.Node('mesh', container * tree.placement, tree.mesh))
  if subs:
    table = tree.mesh[3] if len(tree.mesh) > 3 else None
    if table is None:
      raise geometry.UnsupportedGeometry(f'{obj.Name}: {part.Name} is a plain mesh without faces; select the whole body')
    keep = []
    for name in subs:
      if not name.startswith('Face') or not 1 <= int(name[4:]) <= len(table):
        raise geometry.UnsupportedGeometry(f'{obj.Name}: {part.Name} has no sub-element {name!r} ({len(table)} faces)')
      f = table[int(name[4:]) - 1]
      keep.append(np.arange(f.first, f.first + f.count))
    tri = tri[np.concatenate(keep)]
  a, b, c = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
  area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
  corners = np.concatenate([a, b, c], axis=1)
  normals = None if vn is None else np.concatenate([vn[tri[:, 0]], vn[tri[:, 1]], vn[tri[:, 2]]], axis=1)
  return corners, normals, area


def bakeSurfaceSource(doc, obj):
  prims, faces, facets = [], [], []
  for part, subs in obj._props.get('ActiveSurfaces') or []:
    own = part.Placement if part.hasProperty('Placement') else None
    for gp in _bake.globalPlacements(doc, part):
      container = gp * own.inverse() if own is not None else gp     # solids_of() applies part.Placement itself
      # (named faces of a BRep import are faces of its stored shape: facets, also where the solid
      #  itself is traced as exact CSG)
      for tree in geometry.solids_of(part, brepFacets=bool(subs)):
        if tree.op == 'mesh':
          facets.append(_meshFacets(obj, part, tree, container, subs))
          continue
        flat = geometry.flatten(tree, container)
        base = len(prims)
        for k, fp in enumerate(flat):
          fp.index = base + k
        prims.extend(flat)
        if subs:
          if len(flat) != 1:
            raise geometry.UnsupportedGeometry(
                f'{obj.Name}: faces {subs} of the boolean result {part.Name} are numbered by OpenCASCADE; '
                f'select the whole body or faces of primitive solids')
          for name in subs:
            faces.append((flat[0], _faceIndex(flat[0].kind, flat[0].params, name)))
        else:
          for fp in flat:
            for f in range(geometry.N_FACES[fp.kind]):
              if (fp.facemask >> f) & 1 and faceArea(fp.kind, fp.params, f) > 0:
                faces.append((fp, f))
  if not faces and not facets:
    raise ValueError(f'surface source {obj.Name} has no ActiveSurfaces selected for emission')
  cond_off, cond_prim, cond_inside = [0], [], []
  for fp in prims:
    for other, inside in fp.conds:
      cond_prim.append(other.index)
      cond_inside.append(1 if inside else 0)
    cond_off.append(len(cond_prim))
  srv = distributions.ScalarRandomVariable(
      **point_source.rvArgs(obj, obj.PowerDensity, variableDomain=point_source.parsedDomain(
          obj._props.get('ThetaDomain', '0, pi/4')), scalarRandomVar=True))
  t_edges, t_cdf = srv.tables()
  obj._props['RandomNumberGeneratorMode'] = srv.mode()
  n = len(prims)
  prim_type = np.array([p.kind for p in prims], dtype=np.int32)
  prim_flags = np.array([1 if p.flip else 0 for p in prims], dtype=np.int32)
  prim_xform = np.array([p.to_world.inverse().rows12() for p in prims], dtype=np.float64).reshape(n, 12)
  prim_params = np.array([p.params for p in prims], dtype=np.float64).reshape(n, 4)
  face_prim = np.array([fp.index for fp, _ in faces], dtype=np.int32)
  face_id = np.array([f for _, f in faces], dtype=np.int32)
  face_area = np.array([faceArea(fp.kind, fp.params, f) for fp, f in faces], dtype=np.float64)
  tri_normals = None
  if facets:
    # one TRIANGLE row + one face per facet, after the analytic primitives (whose condition indices stay valid)
    corners = np.concatenate([c for c, _, _ in facets])
    k = len(corners)
    smooth = all(nrm is not None for _, nrm, _ in facets)
    if smooth:
      tri_normals = np.concatenate([np.zeros((n, 9))] + [nrm for _, nrm, _ in facets])
    prim_type = np.concatenate([prim_type, np.full(k, geometry.TRIANGLE, dtype=np.int32)])
    prim_flags = np.concatenate([prim_flags, np.zeros(k, dtype=np.int32)])
    prim_xform = np.concatenate([prim_xform, np.hstack([corners, np.zeros((k, 3))])])
    prim_params = np.concatenate([prim_params, np.zeros((k, 4))])
    cond_off = cond_off + [cond_off[-1]] * k
    face_prim = np.concatenate([face_prim, np.arange(n, n + k, dtype=np.int32)])
    face_id = np.concatenate([face_id, np.zeros(k, dtype=np.int32)])
    face_area = np.concatenate([face_area, np.concatenate([a for _, _, a in facets])])
  return BakedSurfaceSource(
      wavelength=float(obj._props.get('Wavelength', 500)), power=1.0, dist_tol=distTol(doc),
      prim_type=prim_type, prim_flags=prim_flags, prim_xform=prim_xform, prim_params=prim_params,
      prim_cond_off=np.array(cond_off, dtype=np.int32), cond_prim=np.array(cond_prim, dtype=np.int32),
      cond_inside=np.array(cond_inside, dtype=np.int32),
      face_prim=face_prim, face_id=face_id, face_area=face_area,
      t_edges=t_edges, t_cdf=t_cdf, name=obj.Name, label=obj._props.get('Label', obj.Name),
      rays_per_iteration_scale=float(obj._props.get('RaysPerIterationScale', 1)),
      prim_to_world=[p.to_world for p in prims], tri_normals=tri_normals)
