"""`Hits` on rows that stay in HBM.

The reference loads every hit into host arrays and bins them with numpy
(`Hits.histogram`, jupyter_utils/hits.py:176-193 -> `Histogram`,
jupyter_utils/histogram.py:24-57).  At 1e7 - 1e8 recorded hits that is a
0.6 - 6.4 GB copy over PCIe per histogram.  `DeviceHits` offers the same
`histogram(...)`, `detectPlaneNormal(...)`, `planeProject3dPoints`-free
interface on the tracer's hit list where it is: the plane search runs on the
host on the reference's own thinned sample (<= 300 rows, fetched), projection,
medians, ranges and binning run on the device (csrc/odw_posthoc.hip) with
numpy's rules, so the `Histogram` that comes back holds the same plane, origin,
edges and counts as `Hits(all rows).histogram(...)`.
"""
import ctypes as C

import numpy as np

from .. import _native
from ..jupyter_utils import hits as _hits
from ..jupyter_utils.histogram import Histogram, _radius_bins

_THIN = 300


_ARANGE10 = np.arange(10.0)


def _linspace10(start, stop):
  """numpy.linspace(start, stop, 10), its arithmetic without its argument handling (numpy/_core/function_base.py:
  arange(num) * ((stop - start) / (num - 1)) + start, the last element = stop): a quarter of the time, the same bits
  (tests/test_plane_screen.py)"""
  step = (stop - start) / 9
  if not (step != 0 and np.isfinite(step)):
    return np.linspace(start, stop, 10)
  y = _ARANGE10 * step
  y += start
  y[-1] = stop
  return y


class DeviceHits:

  def __init__(self, tracer, group=None):
    """group: index or name of the recording group (None: every group's rows, like
    `loadHits('*')`)"""
    self._tr = tracer
    if isinstance(group, str):
      group = tracer.scene.group_index(group)
    self._group = -1 if group is None else int(group)
    n, leaving = C.c_uint64(0), C.c_uint64(0)
    tracer._chk(tracer._lib.odw_hits_select(tracer._ctx, C.c_int32(self._group), C.byref(n), C.byref(leaving)),
                'odw_hits_select')
    self._n, self._leaving = int(n.value), int(leaving.value)

  def __len__(self):
    return self._n

  # -- small fetches ---------------------------------------------------------
  def _gather(self, entering_only, stride):
    tr = self._tr
    n = C.c_uint64(0)
    tr._chk(tr._lib.odw_hits_gather(tr._ctx, C.c_int32(1 if entering_only else 0), C.c_uint64(stride), None,
                                    C.c_uint64(0), C.byref(n)), 'odw_hits_gather')
    out = np.zeros(int(n.value), dtype=_native.HIT_DTYPE)
    if len(out):
      tr._chk(tr._lib.odw_hits_gather(tr._ctx, C.c_int32(1 if entering_only else 0), C.c_uint64(stride),
                                      out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(n)), 'odw_hits_gather')
    return out

  def _sample(self, limit=_THIN):
    """the rows detectPlaneNormal looks at (hits.py:108-113): points[::k], and directions[::k'] of
    the entering rows only unless leaving rows are the majority"""
    pts = self._gather(False, 1 + int(self._n / limit))['point']
    if self._leaving < .51 * self._n:
      m = self._n - self._leaving
      dirs = self._gather(True, 1 + int(m / limit))['direction']
    else:
      dirs = self._gather(False, 1 + int(self._n / limit))['direction']
    return pts, dirs

  # -- the Hits interface ------------------------------------------------------
  def detectPlaneNormal(self, planeNormal=None, xInPlaneVec=None, maxPointCountConsidered=_THIN, angleTol=1e-9):
    cloud, rays = self._sample(maxPointCountConsidered)
    if planeNormal is None:
      planeNormal = self._flattest_direction(cloud, angleTol)
    planeNormal = _hits._against(planeNormal, rays)
    return planeNormal, _hits._in_plane_x(planeNormal, xInPlaneVec)

  def _flattest_direction(self, cloud, angleTol):
    """the plane search of the host `Hits` class with the screen of every grid run by the library
    (`odw_plane_screen`, outside the interpreter lock -- in a parameter sweep the measuring threads and the baking
    thread share that lock, and the search was half of a thread's time under it); candidates within rounding of the
    smallest extent are evaluated the reference's way, as there: same winner bit for bit"""
    cloud = np.ascontiguousarray(cloud, dtype=np.float64)
    if cloud.ndim != 2 or cloud.shape[1] != 3 or not len(cloud):
      return _hits._flattest_direction(cloud, angleTol)
    lib = self._tr._lib
    pd = C.POINTER(C.c_double)
    cloud_p, n = cloud.ctypes.data_as(pd), C.c_uint64(len(cloud))
    margin = 1e-12 * max(float(np.abs(cloud).max()), 1e-300)
    phis, thetas = np.linspace(0, np.pi, 30), np.linspace(-np.pi / 2, np.pi / 2, 30)
    while True:
      cell = (phis[1] - phis[0], thetas[1] - thetas[0])
      rough = np.empty(len(phis) * len(thetas))
      if lib.odw_plane_screen(cloud_p, n, phis.ctypes.data_as(pd), C.c_int32(len(phis)), thetas.ctypes.data_as(pd),
                              C.c_int32(len(thetas)), rough.ctypes.data_as(pd)) != 0:
        return _hits._flattest_direction(cloud, angleTol)
      near = np.flatnonzero(rough <= rough.min() + margin)
      if len(near) == 0:                     # (a cloud with a nan: the host routine's business)
        return _hits._flattest_direction(cloud, angleTol)
      k = int(near[0])
      if len(near) > 1:                      # (one candidate clear of the others: the reference's first minimum is that one)
        cp, sp, ct, st = np.cos(phis), np.sin(phis), np.cos(thetas), np.sin(thetas)
        best = None
        for c in near:
          i, j = divmod(int(c), len(phis))
          along = np.dot(cloud, np.array([cp[j] * st[i], sp[j] * st[i], ct[i]]))
          e = along.max() - along.min()
          if best is None or e < best:
            best, k = e, int(c)
      i, j = divmod(k, len(phis))
      p, t = phis[j], thetas[i]
      phis = _linspace10(p - 1.1 * cell[0], p + 1.1 * cell[0])
      thetas = _linspace10(t - 1.1 * cell[1], t + 1.1 * cell[1])
      if max(phis[1] - phis[0], thetas[1] - thetas[0]) < angleTol:
        return np.array([np.cos(p) * np.sin(t), np.sin(p) * np.sin(t), np.cos(t)])

  def histogram(self, planeNormal=None, xInPlaneVec=None, key='points', origin=None, radius=None,
                binCoords='cartesian', **kwargs):
    if key not in ('points', 'directions'):
      raise ValueError('DeviceHits.histogram bins points or directions')
    if self._n == 0:
      raise ValueError('no hits to bin')
    if planeNormal is None or xInPlaneVec is None:
      planeNormal, xInPlaneVec = self.detectPlaneNormal(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec)
    ex = np.asarray(xInPlaneVec, dtype=np.float64)
    ey = np.cross(planeNormal, xInPlaneVec)
    ex, ey = ex / np.linalg.norm(ex), ey / np.linalg.norm(ey)
    tr = self._tr
    pd = C.POINTER(C.c_double)
    stats = np.zeros(8)
    tr._chk(tr._lib.odw_hits_project(tr._ctx, C.c_int32(0 if key == 'points' else 1), ex.ctypes.data_as(pd),
                                     ey.ctypes.data_as(pd), stats.ctypes.data_as(pd)), 'odw_hits_project')
    if origin is None:
      # numpy.median: the middle element, or the mean of the two middle ones
      origin = np.array([np.mean(stats[0:2]), np.mean(stats[4:6])])
    origin = np.asarray(origin, dtype=np.float64)
    mode = binCoords.lower()
    if mode in 'cartesian':
      polar = False
    elif mode in 'polar':
      polar = True
    else:
      raise ValueError(f'found invalid binCoord mode {binCoords!r}, expect one of "cartesian" or "polar"')
    if radius is not None:
      _radius_bins(kwargs, radius, polar=polar)
    bins = kwargs.pop('bins', 10)
    if kwargs:
      raise TypeError(f'DeviceHits.histogram: unsupported arguments {sorted(kwargs)}')
    edges = self._edges(bins, polar, origin)
    counts = np.zeros((len(edges[0]) - 1) * (len(edges[1]) - 1), dtype=np.uint64)
    tr._chk(tr._lib.odw_hits_bin(tr._ctx, C.c_int32(1 if polar else 0), origin.ctypes.data_as(pd),
                                 edges[0].ctypes.data_as(pd), C.c_int32(len(edges[0])), edges[1].ctypes.data_as(pd),
                                 C.c_int32(len(edges[1])), counts.ctypes.data_as(C.POINTER(C.c_uint64))), 'odw_hits_bin')
    hist = counts.reshape(len(edges[0]) - 1, len(edges[1]) - 1).astype(np.float64)
    return Histogram.fromBinned(hist, edges[0], edges[1], planeNormal, xInPlaneVec, origin,
                                'polar' if polar else 'cartesian')

  def _edges(self, bins, polar, origin):
    """numpy.histogram2d's reading of `bins` (numpy/lib/_twodim_base_impl.py, _histograms_impl.py):
    one array = the same edges for both coordinates; an integer = that many equal bins over the
    data's range (a degenerate range is widened by 0.5 to both sides)"""
    try:
      n = len(bins)
    except TypeError:
      n = 1
    if n != 1 and n != 2:
      bins = [np.asarray(bins), np.asarray(bins)]
    elif n == 1:
      bins = [bins, bins] if np.ndim(bins) == 0 else [bins[0], bins[0]]
    rng = None
    out = []
    for i, b in enumerate(bins):
      if np.ndim(b) == 0:
        if rng is None:
          rng = np.zeros(4)
          pd = C.POINTER(C.c_double)
          self._tr._chk(self._tr._lib.odw_hits_range(self._tr._ctx, C.c_int32(1 if polar else 0),
                                                     origin.ctypes.data_as(pd), rng.ctypes.data_as(pd)), 'odw_hits_range')
        lo, hi = rng[2 * i], rng[2 * i + 1]
        if lo == hi:
          lo, hi = lo - 0.5, hi + 0.5
        if int(b) < 1:
          raise ValueError('`bins` must be positive, when an integer')
        out.append(np.linspace(lo, hi, int(b) + 1))
      else:
        e = np.ascontiguousarray(b, dtype=np.float64)
        if e.ndim != 1 or len(e) < 2 or np.any(e[:-1] > e[1:]):
          raise ValueError('`bins` must be 1d and increase monotonically, when an array')
        out.append(e)
    return out

  def moments(self):
    """(mean (3,), variance about it (3,)) of the points"""
    tr = self._tr
    pd = C.POINTER(C.c_double)
    mean, var = np.zeros(3), np.zeros(3)
    tr._chk(tr._lib.odw_hits_moments(tr._ctx, mean.ctypes.data_as(pd), var.ctypes.data_as(pd)), 'odw_hits_moments')
    return mean, var

  def rmsSpot(self):
    """rms distance of the hits from their centroid"""
    return float(np.sqrt(self.moments()[1].sum()))

  # -- full copies, for whoever needs the arrays after all ----------------------
  def toHits(self):
    """every selected row on the host, as the reference's `Hits`"""
    rows = self._gather(False, 1)
    tags = rows['tag']
    return _hits.Hits(dict(points=np.ascontiguousarray(rows['point']), directions=np.ascontiguousarray(rows['direction']),
                           powers=np.ascontiguousarray(rows['power']), isEntering=(tags >> np.uint64(63)).astype(np.int64)))

  def points(self):
    return self.toHits().points()

  def directions(self):
    return self.toHits().directions()

  def isEntering(self):
    return self.toHits().isEntering()
