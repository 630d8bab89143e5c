"""Hit point clouds and detector binning on the host.

Same interface and results as the reference's `Hits`
(jupyter_utils/hits.py:21-193): `points()`, `directions()`, `isEntering()`,
`detectPlaneNormal`, `planeProject3dPoints`, `histogram`.  The plane search is
the reference's grid refinement (30x30 start grid, 10x10 refinements around
the best cell until the step is below 1e-9 rad) evaluated as one matrix
product per refinement instead of a Python loop per candidate.
Plot helpers and fan math are not part of the accelerated path.
"""
import warnings

import numpy as np

from .histogram import Histogram

_UNIT = np.eye(3)


def _sphere_dirs(phis, thetas):
  # candidate order of the reference: meshgrid(phis, thetas) flattened row-major
  P, T = np.meshgrid(phis, thetas)
  P, T = P.ravel(), T.ravel()
  return P, T, np.stack([np.cos(P) * np.sin(T), np.sin(P) * np.sin(T), np.cos(T)], axis=1)


class Hits:
  '''
  Class representing a hit coordinate point cloud.
  '''

  def __init__(self, hits):
    self.hits = hits

  def __len__(self):
    return len(self.points())

  def __iter__(self):
    return iter(self.hits.keys())

  def items(self):
    return self.hits.items()

  def keys(self):
    return self.hits.keys()

  def values(self):
    return self.hits.values()

  def _get(self, key):
    return self.hits[key] if key in self.hits else np.array([])

  def points(self):
    return self._get('points')

  def directions(self):
    return self._get('directions')

  def isEntering(self):
    return self._get('isEntering')

  # ---------------------------------------------------------------------
  def detectPlaneNormal(self, points=None, directions=None, planeNormal=None, xInPlaneVec=None,
                        maxPointCountConsidered=300, angleTol=1e-9):
    if points is None:
      points = self.points()
    if directions is None:
      directions = self.directions()
      entering = self.isEntering()
      if np.sum(entering == 0) < .51 * len(entering):
        directions = directions[entering != 0]
    pts = points[::1 + int(points.shape[0] / maxPointCountConsidered)]
    drs = directions[::1 + int(directions.shape[0] / maxPointCountConsidered)]

    if planeNormal is None:
      phis, thetas = np.linspace(0, np.pi, 30), np.linspace(-np.pi / 2, np.pi / 2, 30)
      while True:
        dphi, dtheta = phis[1] - phis[0], thetas[1] - thetas[0]
        P, T, normals = _sphere_dirs(phis, thetas)
        proj = normals @ pts.T
        k = int(np.argmin(proj.max(axis=1) - proj.min(axis=1)))
        phis = np.linspace(P[k] - 1.1 * dphi, P[k] + 1.1 * dphi, 10)
        thetas = np.linspace(T[k] - 1.1 * dtheta, T[k] + 1.1 * dtheta, 10)
        if phis[1] - phis[0] < angleTol and thetas[1] - thetas[0] < angleTol:
          planeNormal = np.array([np.cos(P[k]) * np.sin(T[k]), np.sin(P[k]) * np.sin(T[k]), np.cos(T[k])])
          break

    # the normal points against the incoming rays
    along = drs @ planeNormal
    if np.quantile(along, 0.1) > 0:
      planeNormal = -planeNormal
    elif np.quantile(along, 0.9) >= 0:
      if np.quantile(along, 0.5) < 0:
        planeNormal = -planeNormal
      warnings.warn('unsure of result when trying to auto-detect sign of plane normal, '
                    'avoid relying on the sign of the planeNormal')

    axes = [xInPlaneVec] if xInPlaneVec is not None else list(_UNIT)
    crosses = [np.cross(planeNormal, a) for a in axes]
    projY = crosses[int(np.argmax([np.linalg.norm(c) for c in crosses]))]
    xInPlaneVec = np.cross(planeNormal, projY)
    if xInPlaneVec.sum() < 0:
      xInPlaneVec = -xInPlaneVec
    return planeNormal, xInPlaneVec

  def planeProject3dPoints(self, points=None, planeNormal=None, xInPlaneVec=None, returnZ=False):
    if points is None:
      points = self.points()
    if planeNormal is None or xInPlaneVec is None:
      planeNormal, xInPlaneVec = self.detectPlaneNormal(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec)
    ex = xInPlaneVec / np.linalg.norm(xInPlaneVec)
    ey = np.cross(planeNormal, xInPlaneVec)
    ey = ey / np.linalg.norm(ey)
    cols = [np.dot(points, ex), np.dot(points, ey)]
    if returnZ:
      cols.append(np.dot(points, planeNormal / np.linalg.norm(planeNormal)))
    return np.array(cols).T

  def histogram(self, planeNormal=None, xInPlaneVec=None, key='points', **kwargs):
    points = self.hits[key]
    if planeNormal is None or xInPlaneVec is None:
      planeNormal, xInPlaneVec = self.detectPlaneNormal(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec)
    X, Y = self.planeProject3dPoints(points, planeNormal=planeNormal, xInPlaneVec=xInPlaneVec).T
    return Histogram(X, Y, planeNormal=planeNormal, xInPlaneVec=xInPlaneVec, **kwargs)
