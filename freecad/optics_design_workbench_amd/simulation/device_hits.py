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


def flattestDirection(lib, cloud, angleTol=1e-9):
  """the plane search of the host `Hits` class with the screen of every grid run by the library
  (`odw_plane_screen`, outside the interpreter lock -- in a parameter sweep the measuring threads and the baking
  thread share that lock, and the search was half of a thread's time under it); candidates within rounding of the
  smallest extent are evaluated the reference's way, as there: same winner bit for bit"""
  cloud = np.ascontiguousarray(cloud, dtype=np.float64)
  if cloud.ndim != 2 or cloud.shape[1] != 3 or not len(cloud):
    return _hits._flattest_direction(cloud, angleTol)
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



def flattestDirections(lib, clouds, angleTol=1e-9):
  """`flattestDirection` for several clouds at once: the searches advance level by level (their grids have the same sizes
  and shrink alike), every level's screen of all clouds is ONE call into the library, which spreads the clouds over
  its threads (`odw_plane_screen_batch`).  Cloud by cloud the same result as `flattestDirection`, bit for bit."""
  clouds = [np.ascontiguousarray(c, dtype=np.float64) for c in clouds]
  S = len(clouds)
  out = [None] * S
  live = [k for k in range(S) if clouds[k].ndim == 2 and clouds[k].shape[1] == 3 and len(clouds[k])]
  for k in range(S):
    if k not in live:
      out[k] = _hits._flattest_direction(clouds[k], angleTol)
  if not live:
    return out
  pd = C.POINTER(C.c_double)
  ptrs = (pd * len(live))(*[clouds[k].ctypes.data_as(pd) for k in live])
  counts = np.array([len(clouds[k]) for k in live], dtype=np.uint64)
  margin = [1e-12 * max(float(np.abs(clouds[k]).max()), 1e-300) for k in live]
  phis = np.tile(np.linspace(0, np.pi, 30), (len(live), 1))
  thetas = np.tile(np.linspace(-np.pi / 2, np.pi / 2, 30), (len(live), 1))
  while True:
    n_phi, n_theta = phis.shape[1], thetas.shape[1]
    rough = np.empty((len(live), n_phi * n_theta))
    if lib.odw_plane_screen_batch(ptrs, counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(len(live)), phis.ctypes.data_as(pd),
                                  C.c_int32(n_phi), thetas.ctypes.data_as(pd), C.c_int32(n_theta), rough.ctypes.data_as(pd)) != 0:
      for k in live:
        out[k] = flattestDirection(lib, clouds[k], angleTol)
      return out
    # (the bookkeeping of a level for all clouds in a few array operations; clouds whose screen leaves several candidates
    #  within rounding of the smallest extent are looked at one by one, the reference's way)
    marg = np.asarray(margin)
    mask = rough <= (rough.min(axis=1) + marg)[:, None]
    n_near = mask.sum(axis=1)
    c_best = mask.argmax(axis=1)
    for a in np.flatnonzero(n_near != 1):
      k = live[a]
      near = np.flatnonzero(mask[a])
      if len(near) == 0:                     # (a cloud with a nan: the host routine's business)
        out[k] = _hits._flattest_direction(clouds[k], angleTol)
        c_best[a] = 0
        continue
      ph, th = phis[a], thetas[a]
      cp, sp, ct, st = np.cos(ph), np.sin(ph), np.cos(th), np.sin(th)
      best = None
      for c in near:
        i, j = divmod(int(c), n_phi)
        along = np.dot(clouds[k], np.array([cp[j] * st[i], sp[j] * st[i], ct[i]]))
        e = along.max() - along.min()
        if best is None or e < best:
          best, c_best[a] = e, int(c)
    rows_ = np.arange(len(live))
    p, t = phis[rows_, c_best % n_phi], thetas[rows_, c_best // n_phi]
    best_pt = list(zip(p, t))
    new = []
    for centre, grid in ((p, phis), (t, thetas)):
      cl = grid[:, 1] - grid[:, 0]
      start, stop = centre - 1.1 * cl, centre + 1.1 * cl
      step = (stop - start) / 9                                  # (_linspace10, row by row)
      y = _ARANGE10[None, :] * step[:, None]
      y += start[:, None]
      y[:, -1] = stop
      for a in np.flatnonzero(~((step != 0) & np.isfinite(step))):
        y[a] = np.linspace(start[a], stop[a], 10)
      new.append(y)
    new_phis, new_thetas = new
    phis, thetas = new_phis, new_thetas
    # (every cloud's grid has the same steps: the searches end together)
    if max(phis[0, 1] - phis[0, 0], thetas[0, 1] - thetas[0, 0]) < angleTol:
      for a, k in enumerate(live):
        if out[k] is None:
          p, t = best_pt[a]
          out[k] = np.array([np.cos(p) * np.sin(t), np.sin(p) * np.sin(t), np.cos(t)])
      return out


def _polar_flag(binCoords):
  mode = binCoords.lower()
  if mode in 'cartesian':
    return False
  if mode in 'polar':
    return True
  raise ValueError(f'found invalid binCoord mode {binCoords!r}, expect one of "cartesian" or "polar"')


def _two_edge_arrays(bins):
  """`bins` as two arrays of edges, or None (integer bin counts, one array for both: the per-segment route)"""
  try:
    n = len(bins)
  except TypeError:
    return None
  if n != 2 or np.ndim(bins[0]) == 0 or np.ndim(bins[1]) == 0:
    return None
  edges = [np.ascontiguousarray(b, dtype=np.float64) for b in bins]
  for e in edges:
    if e.ndim != 1 or len(e) < 2 or np.any(e[:-1] > e[1:]):
      raise ValueError('`bins` must be 1d and increase monotonically, when an array')
  return edges


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
    return flattestDirection(self._tr._lib, cloud, angleTol)

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

  def thinned(self, count):
    """the rows [::max(1, n // count)] of the selection in (ray, bounce) order (HIT_DTYPE): what `points[::k]` picks
    from the arrays `loadHits()` returns"""
    return self._gather(False, max(1, self._n // int(count)))

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


class DeviceHitsBatch:
  """the segments of a batch launch (`Tracer.traceBatch`), measured together: every step of `DeviceHits.histogram` --
  ordered selection, the thinned sample, [the plane search, on the host, scene by scene], projection + medians +
  moments, binning -- runs for all scenes at once on the device (`odw_batch_hits_*`; every decision between two kernels
  is taken there).  Scene by scene the results are those of `DeviceHits` on that scene's segment, bit for bit.  A scene
  the batch route cannot serve (`ordered[k]` False: a ray with two selected rows, a mixed list of entering and leaving
  rows) yields None and is measured through `Tracer.batchSelect(k)` + `DeviceHits`.

  Two ways to drive it: step by step (the constructor selects and waits; `histograms` / `moments` / `thinned` wait once
  each), or as a chain that is enqueued and polled -- `DeviceHitsBatch.begin(...)`, `sampled()`, `searchPlanes()`,
  `enqueueMeasure(...)`, `measured()` -- so that one host thread keeps several tracers' chains in flight (a parameter sweep)."""

  def __init__(self, tracer, n_scenes, group=None, _begin=None):
    self._tr = tracer
    self._S = int(n_scenes)
    if isinstance(group, str):
      group = tracer.scene.group_index(group)
    self._group = -1 if group is None else int(group)
    self._planes = None          # per scene (planeNormal, xInPlaneVec), automatic choice
    self._projected = None       # (stats [S][8], moments [S][6]) of the automatic planes
    self._sample = None          # (rows [S][cap], counts [S]) of the plane search's sample
    self._request = None         # the histogram the enqueued chain bins: (polar, edges_a, edges_b)
    self._binned = None          # its counts [S][nbins], origins [S][2], flags [S]
    self._stage = None           # 'begun' | 'sampled' | 'measuring' | 'measured' (chains only)
    self._keep, self._kept = 0, None
    pu = C.POINTER(C.c_uint64)
    if _begin is not None:
      tracer._chk(tracer._lib.odw_batch_hits_begin(tracer._ctx, C.c_int32(self._group), C.c_uint64(int(_begin))), 'odw_batch_hits_begin')
      self._limit = int(_begin)
      self._stage = 'begun'
      self.rows = self.leaving = self.ordered = None
      return
    n, leaving = np.zeros(self._S, dtype=np.uint64), np.zeros(self._S, dtype=np.uint64)
    ordered = np.zeros(self._S, dtype=np.int32)
    tracer._chk(tracer._lib.odw_batch_hits_select(tracer._ctx, C.c_int32(self._group), n.ctypes.data_as(pu),
                                                  leaving.ctypes.data_as(pu), ordered.ctypes.data_as(C.POINTER(C.c_int32))),
                'odw_batch_hits_select')
    self._note(n, leaving, ordered)

  def _note(self, n, leaving, ordered):
    self.rows = [int(v) for v in n]
    self.leaving = [int(v) for v in leaving]
    self.ordered = [bool(v) and r > 0 for v, r in zip(ordered, self.rows)]

  def __len__(self):
    return self._S

  # -- the chain: enqueue and poll ------------------------------------------------------------------------------------
  @classmethod
  def begin(cls, tracer, n_scenes, group=None, limit=_THIN):
    """enqueue selection + sample behind the tracer's batch launch; nothing is waited for"""
    return cls(tracer, n_scenes, group, _begin=limit)

  def sampled(self, wait=False):
    """True once the sample and the row counts have arrived (wait: block until then)"""
    if self._stage != 'begun':
      return self._stage is not None
    tr = self._tr
    cap = self._limit + 8
    rows = np.zeros((self._S, cap), dtype=_native.HIT_DTYPE)
    n, leaving, counts = (np.zeros(self._S, dtype=np.uint64) for _ in range(3))
    ordered = np.zeros(self._S, dtype=np.int32)
    pu = C.POINTER(C.c_uint64)
    rc = tr._lib.odw_batch_hits_sampled(tr._ctx, C.c_int32(1 if wait else 0), n.ctypes.data_as(pu), leaving.ctypes.data_as(pu),
                                        ordered.ctypes.data_as(C.POINTER(C.c_int32)), rows.ctypes.data_as(C.c_void_p), C.c_uint64(cap),
                                        counts.ctypes.data_as(pu))
    if rc == _native.BUSY:
      return False
    tr._chk(rc, 'odw_batch_hits_sampled')
    self._note(n, leaving, ordered)
    self._sample = (rows, counts)
    self._stage = 'sampled'
    return True

  def enqueueMeasure(self, binCoords='cartesian', bins=None, keep=0):
    """enqueue projection, medians, moments and the histogram `bins` (two edge arrays) about the median origin for the
    automatic planes; nothing is waited for.  bins None: a one-bin histogram (moments only)"""
    polar = _polar_flag(binCoords)
    edges = _two_edge_arrays(bins) if bins is not None else [np.array([-np.inf, np.inf]), np.array([-np.inf, np.inf])]
    if edges is None:
      raise ValueError('DeviceHitsBatch.enqueueMeasure bins two arrays of edges')
    ex, ey, skip = self._axes()
    tr = self._tr
    pd = C.POINTER(C.c_double)
    tr._chk(tr._lib.odw_batch_hits_measure(tr._ctx, ex.ctypes.data_as(pd), ey.ctypes.data_as(pd), skip.ctypes.data_as(C.POINTER(C.c_int32)),
                                           C.c_int32(1 if polar else 0), edges[0].ctypes.data_as(pd), C.c_int32(len(edges[0])),
                                           edges[1].ctypes.data_as(pd), C.c_int32(len(edges[1])), C.c_uint64(int(keep or 0))), 'odw_batch_hits_measure')
    self._keep = int(keep or 0)
    self._request = (polar, edges[0], edges[1])
    self._stage = 'measuring'

  def measured(self, wait=False):
    """True once the enqueued measure has arrived"""
    if self._stage != 'measuring':
      return self._stage == 'measured'
    tr = self._tr
    polar, ea, eb = self._request
    nb = (len(ea) - 1) * (len(eb) - 1)
    stats, moments, origins = np.zeros((self._S, 8)), np.zeros((self._S, 6)), np.zeros((self._S, 2))
    counts, flags = np.zeros((self._S, nb), dtype=np.uint64), np.zeros(self._S, dtype=np.uint32)
    pd = C.POINTER(C.c_double)
    keep_cap = 2 * self._keep + 8 if self._keep else 1
    keep_rows = np.zeros((self._S, keep_cap), dtype=_native.HIT_DTYPE)
    n_keep = np.zeros(self._S, dtype=np.uint64)
    rc = tr._lib.odw_batch_hits_measured(tr._ctx, C.c_int32(1 if wait else 0), stats.ctypes.data_as(pd), moments.ctypes.data_as(pd),
                                         origins.ctypes.data_as(pd), counts.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         flags.ctypes.data_as(C.POINTER(C.c_uint32)), keep_rows.ctypes.data_as(C.c_void_p),
                                         C.c_uint64(keep_cap), n_keep.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc == _native.BUSY:
      return False
    tr._chk(rc, 'odw_batch_hits_measured')
    # (a scene the device could not finish -- more median candidates than it ranks -- is the caller's, per segment)
    for k in np.flatnonzero(flags):
      self._planes[k] = None
    self._projected = (stats, moments)
    self._binned = (counts, origins, flags)
    if self._keep:
      self._kept = (self._keep, [keep_rows[k, :int(n_keep[k])].copy() if self.ordered[k] else None for k in range(self._S)])
    self._stage = 'measured'
    return True

  def detached(self):
    """True when every answer the chain was asked for is on the host and no scene was left to the per-segment calls: the
    context may trace its next batch while the caller still reads this one (histograms of the chain's request, moments,
    the kept sample)"""
    return (self._stage == 'measured' and self._binned is not None and self._projected is not None and
            not np.any(self._binned[2]) and all(self.ordered) and self._planes is not None and
            all(p is not None for p in self._planes) and (not self._keep or getattr(self, '_kept', None) is not None))

  # -- steps ----------------------------------------------------------------------------------------------------------
  def _sampleRows(self):
    if self._sample is None:
      tr = self._tr
      cap = _THIN + 8
      rows = np.zeros((self._S, cap), dtype=_native.HIT_DTYPE)
      counts = np.zeros(self._S, dtype=np.uint64)
      tr._chk(tr._lib.odw_batch_hits_sample(tr._ctx, C.c_uint64(_THIN), None, rows.ctypes.data_as(C.c_void_p), C.c_uint64(cap),
                                            counts.ctypes.data_as(C.POINTER(C.c_uint64))), 'odw_batch_hits_sample')
      self._sample = (rows, counts)
    return self._sample

  def _clouds(self):
    rows, counts = self._sampleRows()
    return [rows[k, :int(counts[k])]['point'] for k in range(self._S) if self.ordered[k]]

  def _setNormals(self, normals):
    rows, counts = self._sampleRows()
    planes, it = [], iter(normals)
    for k in range(self._S):
      if not self.ordered[k]:
        planes.append(None)
        continue
      normal = _hits._against(next(it), rows[k, :int(counts[k])]['direction'])
      planes.append((normal, _hits._in_plane_x(normal, None)))
    self._planes = planes

  def _detectPlanes(self):
    if self._planes is None:
      self._setNormals(flattestDirections(self._tr._lib, self._clouds()))
    return self._planes

  searchPlanes = _detectPlanes

  @staticmethod
  def searchPlanesTogether(batches):
    """the plane searches of several batches in lockstep (one screen call per level for all their scenes)"""
    batches = [b for b in batches if b._planes is None]
    if not batches:
      return
    clouds = [b._clouds() for b in batches]
    normals = iter(flattestDirections(batches[0]._tr._lib, [c for cs in clouds for c in cs]))
    for b, cs in zip(batches, clouds):
      b._setNormals([next(normals) for _ in cs])

  def _axes(self):
    planes = self._detectPlanes()
    ex, ey = np.zeros((self._S, 3)), np.zeros((self._S, 3))
    skip = np.ones(self._S, dtype=np.int32)
    for k, pl in enumerate(planes):
      if pl is None:
        continue
      normal, xvec = pl
      x = np.asarray(xvec, dtype=np.float64)
      y = np.cross(normal, xvec)
      ex[k], ey[k] = x / np.linalg.norm(x), y / np.linalg.norm(y)
      skip[k] = 0
    return ex, ey, skip

  def _project(self):
    if self._projected is None:
      tr = self._tr
      ex, ey, skip = self._axes()
      stats, moments = np.zeros((self._S, 8)), np.zeros((self._S, 6))
      pd = C.POINTER(C.c_double)
      tr._chk(tr._lib.odw_batch_hits_project(tr._ctx, ex.ctypes.data_as(pd), ey.ctypes.data_as(pd),
                                             skip.ctypes.data_as(C.POINTER(C.c_int32)), stats.ctypes.data_as(pd),
                                             moments.ctypes.data_as(pd)), 'odw_batch_hits_project')
      self._projected = (stats, moments)
    return self._projected

  def histograms(self, binCoords='cartesian', bins=10, **kwargs):
    """per scene `DeviceHits.histogram(binCoords=..., bins=...)` with the automatic plane and the median origin, or None
    where the scene (or the request: integer bin counts, `radius`, a given plane or origin, directions) takes the
    per-segment route"""
    if kwargs:
      return [None] * self._S
    polar = _polar_flag(binCoords)
    edges = _two_edge_arrays(bins)
    if edges is None:
      return [None] * self._S
    planes = self._detectPlanes()
    req = self._request
    if (self._binned is not None and req[0] == polar and np.array_equal(req[1], edges[0]) and np.array_equal(req[2], edges[1])):
      counts, origins, _ = self._binned          # (the chain binned exactly this)
    else:
      stats, _ = self._project()
      origins = np.zeros((self._S, 2))
      for k in range(self._S):
        if planes[k] is not None:
          origins[k] = (np.mean(stats[k, 0:2]), np.mean(stats[k, 4:6]))       # numpy.median: the mean of the two middle ones
      nb = (len(edges[0]) - 1) * (len(edges[1]) - 1)
      counts = np.zeros((self._S, nb), dtype=np.uint64)
      tr = self._tr
      pd = C.POINTER(C.c_double)
      tr._chk(tr._lib.odw_batch_hits_bin(tr._ctx, C.c_int32(1 if polar else 0), origins.ctypes.data_as(pd),
                                         edges[0].ctypes.data_as(pd), C.c_int32(len(edges[0])), edges[1].ctypes.data_as(pd),
                                         C.c_int32(len(edges[1])), counts.ctypes.data_as(C.POINTER(C.c_uint64))), 'odw_batch_hits_bin')
    out = []
    for k in range(self._S):
      if planes[k] is None:
        out.append(None)
        continue
      hist = counts[k].reshape(len(edges[0]) - 1, len(edges[1]) - 1).astype(np.float64)
      out.append(Histogram.fromBinned(hist, edges[0], edges[1], planes[k][0], planes[k][1], origins[k].copy(),
                                      'polar' if polar else 'cartesian'))
    return out

  def thinned(self, count):
    """per scene the rows [::max(1, n // count)] of the selection in (ray, bounce) order -- what `points[::k]` picks from
    the arrays `loadHits()` returns --, as HIT_DTYPE arrays; None where the scene takes the per-segment route"""
    kept = getattr(self, '_kept', None)
    if kept is not None and kept[0] == int(count):
      return kept[1]                 # (the chain brought them along)
    tr = self._tr
    strides = np.array([max(1, r // int(count)) for r in self.rows], dtype=np.uint64)
    cap = max([-(-r // int(st)) for r, st in zip(self.rows, strides)] + [1])
    rows = np.zeros((self._S, cap), dtype=_native.HIT_DTYPE)
    counts = np.zeros(self._S, dtype=np.uint64)
    pu = C.POINTER(C.c_uint64)
    tr._chk(tr._lib.odw_batch_hits_sample(tr._ctx, C.c_uint64(0), strides.ctypes.data_as(pu), rows.ctypes.data_as(C.c_void_p),
                                          C.c_uint64(cap), counts.ctypes.data_as(pu)), 'odw_batch_hits_sample')
    return [rows[k, :int(counts[k])].copy() if self.ordered[k] else None for k in range(self._S)]

  def moments(self):
    """per scene (mean (3,), variance (3,)) of the points, or None"""
    _, mom = self._project()
    planes = self._detectPlanes()
    return [None if planes[k] is None else (mom[k, :3].copy(), mom[k, 3:].copy()) for k in range(self._S)]
