"""Hit point clouds and detector binning on the host.

Same interface and results as the reference's `Hits`
(jupyter_utils/hits.py:21-193): `points()`, `directions()`, `isEntering()`,
`detectPlaneNormal`, `planeProject3dPoints`, `histogram`.  The plane search is
the reference's grid refinement (30x30 start grid, 10x10 refinements around
the best cell until the step is below 1e-9 rad) evaluated as one matrix
product per refinement instead of a Python loop per candidate.
Fan math (jupyter_utils/hits.py:227-444): per-fan ray spacing, curvature,
estimated power density and caustic indicators from the `fanIndex`,
`rayIndex`, `totalRaysInFan` metadata of a fan-mode run -- same numbers as the
reference (tests/golden/fan_math.npz).  Plot helpers are not provided.
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

  def plot(self, hueKey=None, hueLabel=None, planeNormal=None, xInPlaneVec=None, plotKey='points', **kwargs):
    """scatter plot of the hits projected into their plane, optionally coloured by a metadata
    column (jupyter_utils/hits.py:196-222)"""
    import matplotlib.pyplot as plt
    if plotKey not in self.hits:
      return None
    if planeNormal is None or xInPlaneVec is None:
      planeNormal, xInPlaneVec = self.detectPlaneNormal(points=self.hits[plotKey], planeNormal=planeNormal,
                                                        xInPlaneVec=xInPlaneVec)
    X, Y = self.planeProject3dPoints(self.hits[plotKey], planeNormal=planeNormal, xInPlaneVec=xInPlaneVec).T
    kwargs.setdefault('s', 4)
    if hueKey is not None:
      sc = plt.scatter(X, Y, c=np.asarray(self.hits[hueKey]), cmap=kwargs.pop('cmap', 'hsv'), **kwargs)
      plt.colorbar(sc).set_label(hueLabel or hueKey)
    else:
      sc = plt.scatter(X, Y, **kwargs)
    n, p = planeNormal, xInPlaneVec
    plt.title(f'plane normal = [{n[0]:.2f}, {n[1]:.2f}, {n[2]:.2f}],\n'
              f'projected $x$ = [{p[0]:.2f}, {p[1]:.2f}, {p[2]:.2f}]', fontsize=10)
    plt.xlabel('projected $x$')
    plt.ylabel('projected $y$')
    plt.gca().axis('equal')
    plt.gca().set_aspect('equal')
    return sc

  # ====================================================
  # fan math (jupyter_utils/hits.py:227-444)

  def supportsFanMath(self):
    return all(k in self.hits.keys() for k in ('rayIndex', 'fanIndex', 'totalRaysInFan'))

  def _raiseIfNotFanMath(self):
    if not len(self.hits):
      raise ValueError('keys rayIndex, fanIndex and totalRaysInFan must exist in hits dictionary, '
                       'but hits dictionary is empty')
    if not self.supportsFanMath():
      raise ValueError('keys rayIndex, fanIndex and totalRaysInFan must exist in hits dictionary, '
                       'make sure you simulated in fan mode and enabled storing the respective metadata keys '
                       'in the active SimulationSettings')

  def raysPerFan(self):
    self._raiseIfNotFanMath()
    return self.hits['totalRaysInFan'][0]

  def allRayIndices(self, fanI=None):
    rI, fI = self.hits['rayIndex'], self.hits['fanIndex']
    return np.unique(rI[fI == fanI]) if fanI is not None else np.unique(rI)

  def fanCount(self):
    self._raiseIfNotFanMath()
    return len(set(self.hits['fanIndex']))

  def fanCenter(self, **kwargs):
    """projected position of the central ray: ray 0 of each fan, or the
    midpoint of rays +1/-1 where ray 0 is absent; averaged over fans"""
    self._raiseIfNotFanMath()
    rI, fI = self.hits['rayIndex'], self.hits['fanIndex']
    pXY = self.planeProject3dPoints(self.hits['points'], **kwargs)
    centers = []
    for fanI in set(fI):
      mine = fI == fanI
      if 0 in rI[mine]:
        centers.extend(pXY[mine & (rI == 0)])
      elif +1 in rI[mine] and -1 in rI[mine]:
        centers.extend((pXY[mine & (rI == +1)] + pXY[mine & (rI == -1)]) / 2)
    return np.mean(centers, axis=0) if len(centers) else np.array([np.nan, np.nan])

  def _fanGeometry(self, pCenter=None, **kwargs):
    key = ('geometry', None if pCenter is None else tuple(pCenter), tuple(sorted(kwargs.items())))
    cache = self.__dict__.setdefault('_fanCache', {})
    if key in cache:
      return cache[key]
    self._raiseIfNotFanMath()
    rI, fI, trf = self.hits['rayIndex'], self.hits['fanIndex'], self.hits['totalRaysInFan']
    pXY = self.planeProject3dPoints(self.hits['points'], **kwargs)
    healthy = True
    centerDists, neighborDists, curvs = [], [], []
    missing, skipped = 0, 0

    def side_direction(vectors):
      # mean of the vectors, each COLUMN scaled by its norm over all rays
      # (the reference normalises along axis 0, hits.py:289-297)
      if not len(vectors):
        return None
      norms = np.sqrt(np.sum(vectors**2, axis=0))
      norms[norms == 0] = 1
      return np.mean(vectors / norms, axis=0)

    for fanI in sorted(set(fI)):
      mine = fI == fanI
      rays = sorted(set(rI[mine]))
      if pCenter is None:
        pCenter = self.fanCenter()
      pCenter = np.array(pCenter)
      missing += np.mean(trf[mine]) - len(rays)
      skipped += np.sum(np.array(rays[1:]) - np.array(rays[:-1]) - 1)
      pos = side_direction(pXY[mine & (rI > 0)] - pCenter)
      neg = side_direction(pXY[mine & (rI < 0)] - pCenter)
      if pos is None and neg is None:
        pos, neg = np.array([1, 0]), np.array([-1, 0])
      elif pos is None:
        pos = -neg
      elif neg is None:
        neg = -pos
      where = {i: np.mean(pXY[mine & (rI == i)], axis=0) for i in rays}
      # sliding window (next, this, previous) over the sorted ray indices
      for k in range(len(rays) + 2):
        i1 = rays[k] if k < len(rays) else None
        i0 = rays[k - 1] if 1 <= k <= len(rays) else None
        i2 = rays[k - 2] if 2 <= k else None
        if i0 is None:
          continue
        p0 = where[i0]
        if i1 is not None:
          neighborDists.append([fanI, (i0 + i1) / 2, np.sqrt(np.sum((p0 - where[i1])**2))])
        sP, sN = np.dot(p0 - pCenter, pos), np.dot(p0 - pCenter, neg)
        if sP > 0 and sN < 0:
          sign = +1
        elif sP < 0 and sN > 0:
          sign = -1
        else:
          if sN != 0 and sP != 0 and np.exp(abs(np.log(sP / sN))) < 5:
            warnings.warn('unsure about center distance value signs, the fan-hit pattern is probably '
                          f'very asymmetric ({sP:.1e}, {sN:.1e})')
          healthy = False
          sign = np.sign(sP - sN)
        centerDists.append([fanI, i0, np.sqrt(np.sum((p0 - pCenter)**2)) * sign])
        if i1 is not None and i2 is not None:
          (x0, y0), (x1, y1), (x2, y2) = p0, where[i1], where[i2]
          curvs.append([fanI, i0, abs((y2 - y1) * x0 - (x2 - x1) * y0 + x2 * y1 - y2 * x1)
                        / np.sqrt((y2 - y1)**2 + (x2 - x1)**2)])
    res = dict(centerDists=np.array(centerDists), neighborDists=np.array(neighborDists), curvs=np.array(curvs),
               missingRays=missing, skippedRays=skipped, rI=rI, fI=fI, pXY=pXY, trf=trf,
               healthySymmetry=healthy)
    cache[key] = res
    return res

  def fanMissingRays(self):
    return self._fanGeometry()['missingRays']

  def fanSkippedRays(self):
    return self._fanGeometry()['skippedRays']

  def fanCenterDists(self, pCenter=None):
    return self._fanGeometry(pCenter=pCenter)['centerDists'].T

  def fanNeighborDists(self):
    return self._fanGeometry()['neighborDists'].T

  def fanCurvs(self):
    return self._fanGeometry()['curvs'].T

  def _fanPower(self, pCenter=None):
    key = ('power', None if pCenter is None else tuple(pCenter))
    cache = self.__dict__.setdefault('_fanCache', {})
    if key in cache:
      return cache[key]
    if pCenter is None:
      pCenter = self.fanCenter()
    cfI, crI, cdist = self.fanCenterDists(pCenter=pCenter)
    nfI, nrI, ndist = self.fanNeighborDists()
    densities, caustics = {}, {}
    for fanI in sorted(set(nfI)):
      densities[fanI], caustics[fanI] = [], []
      for between in sorted(nrI[fanI == nfI]):
        # the two rays around the half-integer index (.6: around 0 these are -1 and +1)
        r1, r2 = int(round(between - .6)), int(round(between + .6))
        d1 = np.mean(cdist[(fanI == cfI) & (crI == r1)])
        d2 = np.mean(cdist[(fanI == cfI) & (crI == r2)])
        power = 1 / np.mean(ndist[(fanI == nfI) & (nrI == between)])
        if d2 < d1:        # order of the rays reversed: a caustic fold
          caustics[fanI].append([d2, d1, power])
        else:
          densities[fanI].append([np.mean([d1, d2]), power])
    densityFuncs = {i: (lambda pos, _d=np.array(d).T: np.interp(pos, *_d, left=0, right=0))
                    for i, d in densities.items()}
    causticFuncs = {i: (lambda p1, p2, _d=np.array(d): sum(1 + abs(r1 - r2) for r1, r2, p in _d
                                                          if r1 <= max(p1, p2) and min(p1, p2) <= r2))
                    for i, d in caustics.items()}
    res = dict(fanDensities=densities, fanDensityFuncs=densityFuncs, causticIntensities=caustics,
               causticIntensityFuncs=causticFuncs, pCenter=pCenter,
               healthySymmetry=self._fanGeometry()['healthySymmetry'])
    cache[key] = res
    return res

  def fanEstimatedPowerDensities(self, pCenter=None):
    return {i: np.array(d).T for i, d in self._fanPower(pCenter)['fanDensities'].items()}

  def fanEstimatedPowerDensityFuncs(self, pCenter=None):
    return self._fanPower(pCenter)['fanDensityFuncs']

  def fanEstimatedCausticIntensities(self, pCenter=None):
    return {i: np.array(d).T for i, d in self._fanPower(pCenter)['causticIntensities'].items()}

  def fanEstimatedCausticIntensityFuncs(self, pCenter=None):
    return self._fanPower(pCenter)['causticIntensityFuncs']

  def fanSymmetryHealthy(self, pCenter=None):
    """False if the fan hit pattern is too distorted to be read as a (bent)
    line; the other fan estimates are then not trustworthy"""
    return self._fanPower(pCenter)['healthySymmetry']
