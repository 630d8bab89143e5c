"""Convenience: open a document and bake everything a trace needs."""
from dataclasses import dataclass

import numpy as np

from .freecad_elements import point_source
from .scene import bake as _bake
from .scene import geometry as _geometry
from .scene.fcstd import Document


@dataclass
class BakedProject:
  document: Document
  scene: '_bake.BakedScene'
  source: 'point_source.BakedSource'
  limits: '_bake.Limits'
  sourceObject: object


def _bakeLightSource(doc, source):
  from .simulation.simulation_loop import bakeLightSource
  return bakeLightSource(doc, source)


def bakeProject(doc, source=None, **traceKwargs):
  """document (or path) -> BakedProject for its first (or the given) source.
  traceKwargs: maxRayLength, maxIntersections, powerTol, distTol as in
  Ray.traceRay (ray.py:36-38)."""
  if not isinstance(doc, Document):
    doc = Document(doc)
  sources = _bake.lightSources(doc)
  if source is None:
    if not sources:
      raise ValueError('document has no light source')
    source = sources[0]
  elif isinstance(source, str):
    source = doc.getObject(source) or doc.getObjectsByLabel(source)[0]
  return BakedProject(document=doc, scene=_bake.bakeScene(doc, source),
                      source=_bakeLightSource(doc, source),
                      limits=_bake.bakeLimits(doc, source, **traceKwargs), sourceObject=source)


def planeDetector(scene, group, nx=1024, ny=1024, window=None, prim=None, face=None, toward=None):
  """detector window on a planar face of a recording group: by default the
  box face of `group` with the largest area (the reference picks the plane
  post hoc from the point cloud, hits.py:96-174; a device histogram needs it
  up front).  Among equally large faces the one whose outward normal points
  most towards the point `toward` (e.g. the source position) wins.
  -> dict for Tracer.setDetector"""
  gi = scene.group_index(group) if isinstance(group, str) else int(group)
  best = None
  for p in range(scene.n_prims):
    if scene.prim_group[p] != gi or scene.prim_type[p] != _geometry.BOX:
      continue
    if prim is not None and p != prim:
      continue
    size = scene.prim_params[p][:3]
    for f in range(6):
      if face is not None and f != face:
        continue
      a = f >> 1
      b1, b2 = (a + 1) % 3, (a + 2) % 3
      area = size[b1] * size[b2]
      score = 0.0
      if toward is not None:
        tw = scene.prim_to_world[p]
        c = np.array(size) / 2
        nrm = np.zeros(3)
        nrm[a] = 1.0 if f & 1 else -1.0
        c_face = c.copy()
        c_face[a] = size[a] if f & 1 else 0.0
        v = np.asarray(toward, dtype=np.float64) - (tw * c_face)
        score = float((tw.m[:3, :3] @ nrm) @ (v / (np.linalg.norm(v) + 1e-300)))
      if best is None or (area, score) > (best[0], best[3]):
        best = (area, p, f, score)
  if best is None:
    raise ValueError(f'group {group} has no planar box face')
  _, p, f, _ = best
  a = f >> 1
  b1, b2 = (a + 1) % 3, (a + 2) % 3
  size = scene.prim_params[p][:3]
  tw = scene.prim_to_world[p]
  centre = np.zeros(3)
  centre[a] = size[a] if f & 1 else 0.0
  centre[b1], centre[b2] = size[b1] / 2, size[b2] / 2
  e1, e2 = np.zeros(3), np.zeros(3)
  e1[b1] = 1.0
  e2[b2] = 1.0
  R = tw.m[:3, :3]
  hx, hy = (size[b1] / 2, size[b2] / 2) if window is None else (window, window)
  return dict(group=gi, origin=(tw * centre).tolist(), ex=(R @ e1).tolist(), ey=(R @ e2).tolist(),
              x_lo=-hx, x_hi=hx, y_lo=-hy, y_hi=hy, nx=int(nx), ny=int(ny))
