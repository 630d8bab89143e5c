"""Hit point clouds and detector binning on the host.

Same interface and results as the reference's `Hits`
(jupyter_utils/hits.py:21-193): `points()`, `directions()`, `isEntering()`,
`detectPlaneNormal`, `planeProject3dPoints`, `histogram`.  The plane search is
the reference's grid refinement (30x30 start grid, 10x10 refinements around
the best cell until the step is below 1e-9 rad), candidates generated per grid
and evaluated one matrix-vector product each, so that plane, origin and counts
equal the reference's bit for bit (tests/golden/hist_cases.npz).
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


def _thinned(a, limit):
  """every k-th row, k chosen so that about `limit` rows remain"""
  return a[::1 + int(a.shape[0] / limit)]


def _incoming_only(directions, entering):
  """without the rows of rays leaving a transparent detector, unless leaving rays are the majority
  (steadies the sign of the plane normal)"""
  if np.sum(entering == 0) < .51 * len(entering):
    return directions[entering != 0]
  return directions


def _flattest_direction(cloud, angleTol):
  """direction along which the cloud has the smallest extent: 30 x 30 grid over half the unit sphere
  (a plane has two normals), then 10 x 10 grids 1.1 cells around the best one until the cells are
  smaller than angleTol"""
  phis, thetas = np.linspace(0, np.pi, 30), np.linspace(-np.pi / 2, np.pi / 2, 30)
  cloud_t = np.ascontiguousarray(np.asarray(cloud, dtype=np.float64).T)
  while True:
    cell = (phis[1] - phis[0], thetas[1] - thetas[0])
    P, T, normals = _sphere_dirs(phis, thetas)
    # The reference evaluates one matrix-vector product per candidate (hits.py:124-128) and takes the
    # first minimum; one matrix-matrix product sums in another order, and the argmin over nearly
    # equal extents then lands on a neighbouring cell now and then.  So: screen all candidates with
    # one vectorised pass, then redo those within rounding of the smallest extent the
    # reference's way -- same winner bit for bit, a fiftieth of the calls.
    # (the screen only has to be right to ~1e-13 of the cloud's size -- the margin below --, so one
    #  matrix product does: 0.5 ms for the 900 candidates of the first grid against 2.9 ms element-wise;
    #  64 x 8 ms of plane search were the larger half of a 64-radius sweep)
    rough = normals @ cloud_t
    rough = rough.max(axis=1) - rough.min(axis=1)
    near = np.flatnonzero(rough <= rough.min() + 1e-12 * max(float(np.abs(cloud).max()), 1e-300))
    exact = np.empty(len(near))
    for j, c in enumerate(near):
      along = np.dot(cloud, normals[c])
      exact[j] = along.max() - along.min()
    k = int(near[int(np.argmin(exact))])
    phis = np.linspace(P[k] - 1.1 * cell[0], P[k] + 1.1 * cell[0], 10)
    thetas = np.linspace(T[k] - 1.1 * cell[1], T[k] + 1.1 * cell[1], 10)
    if max(phis[1] - phis[0], thetas[1] - thetas[0]) < angleTol:
      return np.array([np.cos(P[k]) * np.sin(T[k]), np.sin(P[k]) * np.sin(T[k]), np.cos(T[k])])


def _signed_quantile(a, q):
  """a number with the sign of numpy.quantile(a, q) -- all `_against` looks at: the quantile interpolates between two
  neighbours of the sorted values, so where the values around that position lie on one side of zero (the usual
  case) one of them will do (numpy.quantile costs 40 us a call, three calls per plane search); where they straddle
  zero, numpy.quantile itself"""
  n = len(a)
  if n >= 4:
    s = np.sort(a)
    v = q * (n - 1)
    below, above = s[max(int(v) - 1, 0)], s[min(int(v) + 2, n - 1)]
    if below > 0:
      return below
    if above < 0:
      return above
  return np.quantile(a, q)


def _against(normal, rays):
  """the normal or its opposite, whichever the rays run against: clear if 90 % of them agree,
  else by the median, with a warning"""
  along = rays @ normal
  lo, mid, hi = _signed_quantile(along, 0.1), _signed_quantile(along, 0.5), _signed_quantile(along, 0.9)
  if lo > 0:
    return -normal
  if hi < 0:
    return normal
  warnings.warn('unsure of result when trying to auto-detect sign of plane normal, '
                'avoid relying on the sign of the planeNormal')
  return -normal if mid < 0 else normal


def _in_plane_x(normal, hint):
  """x axis in the plane: normal x (normal x a), a = the hint or the coordinate axis that gives the
  longest cross product (numerically safest); signed so that its components sum to >= 0"""
  crosses = [np.cross(normal, a) for a in ([hint] if hint is not None else list(_UNIT))]
  longest = crosses[int(np.argmax([np.linalg.norm(c) for c in crosses]))]
  x = np.cross(normal, longest)
  return -x if x.sum() < 0 else x


class Hits:
  '''
  The hits of a simulation run as a point cloud with metadata columns: plane detection and
  projection, histograms, plots, and the fan-mode estimates.
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
    """(unit normal of the plane the hit cloud is flattest along, pointing against the incoming rays;
    in-plane x axis) -- jupyter_utils/hits.py:96-174: nested angular grid search over half the sphere
    on a thinned cloud, sign from the quantiles of direction . normal, x axis from the coordinate
    axis least parallel to the normal (or the caller's hint)"""
    cloud = self.points() if points is None else points
    if directions is None:
      directions = _incoming_only(self.directions(), self.isEntering())
    cloud, rays = _thinned(cloud, maxPointCountConsidered), _thinned(directions, maxPointCountConsidered)
    if planeNormal is None:
      planeNormal = _flattest_direction(cloud, angleTol)
    planeNormal = _against(planeNormal, rays)
    return planeNormal, _in_plane_x(planeNormal, xInPlaneVec)

  def planeProject3dPoints(self, points=None, planeNormal=None, xInPlaneVec=None, returnZ=False):
    """(n, 2) coordinates in the plane's own axes (x, normal x x); returnZ adds the height above it"""
    if points is None:
      points = self.points()
    if planeNormal is None or xInPlaneVec is None:
      planeNormal, xInPlaneVec = self.detectPlaneNormal(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec)
    axes = [xInPlaneVec, np.cross(planeNormal, xInPlaneVec)] + ([planeNormal] if returnZ else [])
    return np.array([np.dot(points, a / np.linalg.norm(a)) for a in axes]).T

  def _flat(self, key, planeNormal, xInPlaneVec, detectOn=None):
    """columns of hits[key] in plane coordinates + the plane used"""
    if planeNormal is None or xInPlaneVec is None:
      planeNormal, xInPlaneVec = self.detectPlaneNormal(points=detectOn, planeNormal=planeNormal, xInPlaneVec=xInPlaneVec)
    xy = self.planeProject3dPoints(self.hits[key], planeNormal=planeNormal, xInPlaneVec=xInPlaneVec)
    return xy[:, 0], xy[:, 1], planeNormal, xInPlaneVec

  def histogram(self, planeNormal=None, xInPlaneVec=None, key='points', **kwargs):
    x, y, n, ex = self._flat(key, planeNormal, xInPlaneVec)
    return Histogram(x, y, planeNormal=n, xInPlaneVec=ex, **kwargs)

  def plot(self, hueKey=None, hueLabel=None, planeNormal=None, xInPlaneVec=None, plotKey='points', **kwargs):
    """scatter plot of the hits in plane coordinates, optionally coloured by a metadata column
    (jupyter_utils/hits.py:196-222); None if there is nothing to plot"""
    if plotKey not in self.hits:
      return None
    import matplotlib.pyplot as plt
    x, y, n, ex = self._flat(plotKey, planeNormal, xInPlaneVec, detectOn=self.hits[plotKey])
    style = dict(s=4)
    style.update(kwargs)
    if hueKey is None:
      dots = plt.scatter(x, y, **style)
    else:
      style.setdefault('cmap', 'hsv')
      dots = plt.scatter(x, y, c=np.asarray(self.hits[hueKey]), **style)
      plt.colorbar(dots).set_label(hueKey if hueLabel is None else hueLabel)
    ax = plt.gca()
    ax.set_title('plane normal = [%.2f, %.2f, %.2f],\nprojected $x$ = [%.2f, %.2f, %.2f]' % (*n, *ex), fontsize=10)
    ax.set_xlabel('projected $x$')
    ax.set_ylabel('projected $y$')
    ax.axis('equal')
    ax.set_aspect('equal')
    return dots

  # ====================================================
  # fan math (jupyter_utils/hits.py:227-444)

  def supportsFanMath(self):
    return all(k in self.hits.keys() for k in ('rayIndex', 'fanIndex', 'totalRaysInFan'))

  def _raiseIfNotFanMath(self):
    if self.supportsFanMath():
      return
    need = 'fan math needs the metadata columns rayIndex, fanIndex and totalRaysInFan'
    if not len(self.hits):
      raise ValueError(need + ', but there are no hits at all')
    raise ValueError(need + ': simulate in fan mode with StoreHitRayIndex, StoreHitFanIndex and '
                     'StoreHitTotalRaysInFan switched on in the active simulation settings')

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
    ray, fan = self.hits['rayIndex'], self.hits['fanIndex']
    xy = self.planeProject3dPoints(self.hits['points'], **kwargs)
    found = []
    for f in set(fan):
      at = {i: xy[(fan == f) & (ray == i)] for i in (0, 1, -1)}
      if len(at[0]):
        found.extend(at[0])
      elif len(at[1]) and len(at[-1]):
        found.extend((at[1] + at[-1]) / 2)
    if not found:
      return np.array([np.nan, np.nan])
    return np.mean(found, axis=0)

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
    c_fan, c_ray, c_dist = self.fanCenterDists(pCenter=pCenter)
    n_fan, n_mid, n_dist = self.fanNeighborDists()

    def radius(f, i):          # signed distance of ray i of fan f from the centre
      return np.mean(c_dist[(c_fan == f) & (c_ray == i)])
    densities, caustics = {}, {}
    for f in sorted(set(n_fan)):
      smooth, folds = [], []
      for mid in sorted(n_mid[n_fan == f]):
        # the gap between two neighbouring rays, labelled by the mean of their indices: the rays are
        # the integers .6 below and above it (around 0 that picks -1 and +1 when ray 0 is missing)
        inner, outer = radius(f, int(round(mid - .6))), radius(f, int(round(mid + .6)))
        power = 1 / np.mean(n_dist[(n_fan == f) & (n_mid == mid)])
        if outer < inner:      # the rays have changed places: a caustic fold
          folds.append([outer, inner, power])
        else:
          smooth.append([np.mean([inner, outer]), power])
      densities[f], caustics[f] = smooth, folds
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
