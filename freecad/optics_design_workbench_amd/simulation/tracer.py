"""Device tracer: the accelerated region of runSimulationIteration.

In the reference `GenericSourceProxy.runSimulationIteration`
(freecad_elements/generic_source.py:51-146) generates RaysPerIteration rays and
walks `ray.traceRay(store=store)` for each; here one `Tracer.trace(first, n,
seed)` call does that for n rays on the GPU through the C-ABI and leaves hit
rows, counters and the detector histogram in HBM until they are fetched.
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from .. import _native
from .._native import CNT_NAMES, HIT_DTYPE, SEGMENT_DTYPE, TRACE_HISTOGRAM, TRACE_RECORD_HITS, TRACE_RECORD_SEGMENTS


# Contexts that are still open when the interpreter ends are closed in ITS order, at the start of its shutdown (atexit),
# while the HIP runtime and every module are whole -- not whenever the garbage collector reaches their Tracer during
# module teardown
_LIVE = weakref.WeakSet()


@atexit.register
def _close_live_tracers():
  for t in list(_LIVE):
    try:
      t.close()
    except Exception:
      pass


class _CudaArrayView:
  """zero-copy view of a device buffer for torch.as_tensor (RCCL reductions)"""

  def __init__(self, ptr, n, typestr, owner):
    self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False),
                                         version=2, strides=None)
    self._owner = owner


class _PinnedPool:
  """page-locked host memory for the arrays a launch hands over (Tracer.hitColumns): the copy engine writes
  into them directly (48 - 56 GB/s; 17 GB/s through the staging copies of pageable memory).  One pool per
  process: the arrays may outlive the tracer that filled them (a run's results kept in memory).  A slab goes
  back to the pool when the last array carved from it is garbage-collected (the run loop's arrays: once the
  writer threads have pickled them), so a continuous run cycles through as many slabs as it has launches in flight
  between the device and the files (writer threads + their queue + the fetch under way: `reserve` tells the pool; with
  fewer slabs kept than that, every launch page-locks a new 300 MB slab and gives one back -- 50 - 100 ms each, which is
  why MORE writer threads made round 4's run loop SLOWER: 4 / 6 / 8 / 12 writers = 2.9 / 2.1 / 1.7 / 1.2e8 rays/s);
  arrays that are kept keep their slab and the pool allocates another; free slabs beyond KEEP are given back to the
  system.  Page-locking costs 0.22 ms per MB on the pool's boxes (67 ms for a launch's 300 MB slab, 55 ms to give it
  back: scripts/bench_pinned_alloc.py) -- fifteen times what the launch itself may take at 5e8 rays/s --, so the slabs stay
  with the process between runs (a second run starts warm); `releasePinnedMemory()` gives them back."""
  KEEP = 4

  def __init__(self):
    import threading
    self._free = []           # (bytes, address)
    self._lock = threading.Lock()

  def _release(self, nbytes, addr):
    drop = None
    with self._lock:
      self._free.append((nbytes, addr))
      if len(self._free) > self.KEEP:
        drop = self._free.pop(0)
    if drop is not None:
      try:
        _native.lib().odw_host_free(None, C.c_void_p(drop[1]))
      except Exception:
        pass

  def reserve(self, slabs):
    """free slabs the pool holds on to from now on (a run's pipeline depth)"""
    self.KEEP = max(4, int(slabs))

  def trim(self, keep=2):
    """after a run: give free slabs beyond `keep` back to the system"""
    self.KEEP = max(int(keep), 0)
    while True:
      with self._lock:
        if len(self._free) <= self.KEEP:
          break
        drop = self._free.pop(0)
      try:
        _native.lib().odw_host_free(None, C.c_void_p(drop[1]))
      except Exception:
        pass
    self.KEEP = 4

  def take(self, tracer, nbytes):
    """a ctypes buffer of >= nbytes page-locked bytes; numpy arrays made from it (np.frombuffer) keep it alive"""
    import weakref
    nbytes = max(int(nbytes), 4096)
    size = addr = None
    with self._lock:
      fit = [k for k, (b, _) in enumerate(self._free) if nbytes <= b <= 2 * nbytes + (1 << 20)]
      if fit:
        size, addr = self._free.pop(min(fit, key=lambda k: self._free[k][0]))
    if addr is None:
      size = (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)
      p = C.c_void_p()
      tracer._chk(tracer._lib.odw_host_alloc(tracer._ctx, C.c_uint64(size), C.byref(p)), 'odw_host_alloc')
      addr = p.value
    buf = (C.c_char * size).from_address(addr)
    weakref.finalize(buf, self._release, size, addr)
    return buf


_POOL = _PinnedPool()


def releasePinnedMemory(keep=0):
  """give the page-locked slabs the run loops of this process keep for their next run back to the system"""
  _POOL.trim(keep)


class Tracer:

  def __init__(self, device=0, referenceStrict=None):
    """referenceStrict: upload scenes without ODW_FLAG_CONVEX, i.e. without the convex-solid skip
    (include/odw_trace.h): every solid of every relevant group is a candidate of every segment, as in
    findNearestIntersection (ray.py:328-364).  Default (None): the environment variable ODW_STRICT, else
    off.  The skip changes results only for rays that leave a convex solid within ~distTol of one of its
    edges (0 of 1e6 rays at DistanceTolerance 1e-6, 2.5e-4 - 5.7e-4 of the rays of the
    two reference scenes with DistanceTolerance 1e-2) and buys 6 % on lensesAndMirrors."""
    import os
    if referenceStrict is None:
      referenceStrict = os.environ.get('ODW_STRICT', '') not in ('', '0')
    self.referenceStrict = bool(referenceStrict)
    self._lib = _native.lib()
    self._ctx = C.c_void_p()
    self.device = int(device)
    _native.check(None, self._lib.odw_create(self.device, C.byref(self._ctx)), 'odw_create')
    _LIVE.add(self)
    self._det = None
    self._keep = {}
    # ODW_COMPILE=structure: every tracer of the process compiles its scenes (test campaigns)
    if os.environ.get('ODW_COMPILE'):
      self.compileScene(os.environ['ODW_COMPILE'])

  # -- lifecycle ------------------------------------------------------------
  def close(self):
    for extra in getattr(self, '_sweepLanes', None) or []:      # (contexts a parameter sweep keeps beside this one)
      extra.close()
    self._sweepLanes = []
    if self._ctx:
      for p in getattr(self, '_pinned', []):
        self._lib.odw_host_free(self._ctx, p)
      self._pinned = []
      self._lib.odw_destroy(self._ctx)
      self._ctx = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    self.close()

  def _chk(self, rc, what):
    _native.check(self._ctx, rc, what)

  # -- uploads --------------------------------------------------------------
  def setScene(self, scene):
    if self.referenceStrict:
      import copy
      stripped = copy.copy(scene)
      stripped.prim_flags = np.asarray(scene.prim_flags, dtype=np.int32) & ~np.int32(_native.FLAG_CONVEX)
      d, keep = _native.scene_desc(stripped)
    else:
      d, keep = _native.scene_desc(scene)
    self._chk(self._lib.odw_upload_scene(self._ctx, C.byref(d)), 'odw_upload_scene')
    arr, n, keep = _native.surface_sampler_descs(getattr(scene, 'surface_samplers', None))
    self._chk(self._lib.odw_upload_surface_samplers(self._ctx, arr, C.c_int32(n)), 'odw_upload_surface_samplers')
    self.scene = scene

  def compileScene(self, mode='structure'):
    """Scene-compiled kernels (odw_compile_scene): 'structure' -- the ray loop compiled against the
    scene's structure when the scene is bound (0.5 - 2 s up to 16 primitives, 17 s for 61; one kernel per
    structure: parameter sweeps reuse it; cached per process and on disk); 'auto' -- never waits: a cached
    kernel is taken at once, otherwise the generic kernels trace, a thread compiles once the scene has
    traced 5e7 rays, and the launch after it has finished switches over; 'off' / None -- the generic
    kernels.  Sticky: applies to the uploaded scene and to every scene set later.  The results are those of
    the generic kernels bit for bit (which is what makes 'auto' invisible); scenes outside the domain keep
    them silently.  Returns compiledInfo()."""
    f = self._lib.odw_compile_scene
    f.argtypes = [C.c_void_p, C.c_int32]
    self._chk(f(self._ctx, _native.COMPILE_MODES[mode]), 'odw_compile_scene')
    self._compileMode = _native.COMPILE_MODES[mode]
    return self.compiledInfo()

  def compileMode(self):
    """the sticky mode compileScene last set: 0 off, 1 structure, 2 auto"""
    return getattr(self, '_compileMode', 0)

  def compiledInfo(self):
    """dict(mode: 0 generic (in 'auto' mode: not compiled yet) / 1 structure / 2 auto -- of the kernel the next eligible launch runs,
    seconds: compile time of it (0 from a cache), cache: 0 compiled now / 1 process / 2 disk)"""
    f = self._lib.odw_compiled_info
    f.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    b, sec, hit = C.c_int32(0), C.c_double(0), C.c_int32(0)
    self._chk(f(self._ctx, C.byref(b), C.byref(sec), C.byref(hit)), 'odw_compiled_info')
    return dict(mode=int(b.value), seconds=float(sec.value), cache=int(hit.value))

  def setSurfaceSeed(self, seed):
    """Philox key of the stochastic-surface draws in traceRays launches"""
    self._chk(self._lib.odw_set_surface_seed(self._ctx, C.c_uint64(int(seed))), 'odw_set_surface_seed')

  def setWavelength(self, wavelength_nm):
    """wavelength of the following launches (explicit rays); setSource resets it"""
    self._chk(self._lib.odw_set_wavelength(self._ctx, C.c_double(float(wavelength_nm))), 'odw_set_wavelength')

  def setSource(self, source):
    """point source (BakedSource) or surface source (BakedSurfaceSource)"""
    if hasattr(source, 'face_prim'):
      d, keep = _native.surface_source_desc(source)
      self._chk(self._lib.odw_upload_surface_source(self._ctx, C.byref(d)), 'odw_upload_surface_source')
    else:
      d, keep = _native.source_desc(source)
      self._chk(self._lib.odw_upload_source(self._ctx, C.byref(d)), 'odw_upload_source')
    self.source = source

  def generateRays(self, first, n, seed):
    """initial conditions only (returnInitialConditions, generic_source.py:57): origins, directions"""
    o = np.empty((int(n), 3))
    d = np.empty((int(n), 3))
    pd = C.POINTER(C.c_double)
    self._chk(self._lib.odw_generate_rays(self._ctx, C.c_uint64(int(first)), C.c_uint64(int(n)),
                                          C.c_uint64(int(seed)), o.ctypes.data_as(pd), d.ctypes.data_as(pd)),
              'odw_generate_rays')
    return o, d

  def setLimits(self, lim):
    d = _native.LimitsDesc(float(lim.max_ray_length), int(lim.max_intersections),
                           float(lim.dist_tol), float(lim.power_tol))
    self._chk(self._lib.odw_set_limits(self._ctx, C.byref(d)), 'odw_set_limits')
    self.limits = lim

  def setDetector(self, det):
    """det: dict(group, origin, ex, ey, x_lo, x_hi, y_lo, y_hi, nx, ny) or None"""
    if det is None:
      self._chk(self._lib.odw_set_detector(self._ctx, None), 'odw_set_detector')
      self._det = None
      return
    d = _native.DetectorDesc()
    d.group = int(det.get('group', -1))
    d.origin = (C.c_double * 3)(*det['origin'])
    d.ex = (C.c_double * 3)(*det['ex'])
    d.ey = (C.c_double * 3)(*det['ey'])
    d.x_lo, d.x_hi, d.y_lo, d.y_hi = (float(det[k]) for k in ('x_lo', 'x_hi', 'y_lo', 'y_hi'))
    d.nx, d.ny = int(det['nx']), int(det['ny'])
    self._chk(self._lib.odw_set_detector(self._ctx, C.byref(d)), 'odw_set_detector')
    self._det = dict(det)

  def reserveHits(self, capacity):
    self._chk(self._lib.odw_reserve_hits(self._ctx, C.c_uint64(int(capacity))), 'odw_reserve_hits')

  def reserveSegments(self, capacity):
    """room for the rows of sources with RecordRays (one per ray segment)"""
    self._chk(self._lib.odw_reserve_segments(self._ctx, C.c_uint64(int(capacity))), 'odw_reserve_segments')

  # -- tracing --------------------------------------------------------------
  @staticmethod
  def _flags(record_hits, histogram, record_segments=False):
    return ((TRACE_RECORD_HITS if record_hits else 0) | (TRACE_HISTOGRAM if histogram else 0)
            | (TRACE_RECORD_SEGMENTS if record_segments else 0))

  def trace(self, first, n, seed, record_hits=True, histogram=True, record_segments=False):
    """asynchronous: rays first..first+n-1 of Philox stream `seed`"""
    self._chk(self._lib.odw_trace(self._ctx, C.c_uint64(int(first)), C.c_uint64(int(n)), C.c_uint64(int(seed)),
                                  C.c_uint32(self._flags(record_hits, histogram, record_segments))),
              'odw_trace')

  def traceRays(self, origins, directions, powers=None, first=0, record_hits=True, histogram=True,
                record_segments=False):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    if o.shape != d.shape:
      raise ValueError('origins and directions differ in shape')
    p = np.ascontiguousarray(powers, dtype=np.float64) if powers is not None else None
    pd = C.POINTER(C.c_double)
    self._chk(self._lib.odw_trace_rays(
        self._ctx, C.c_uint64(int(first)), C.c_uint64(len(o)), o.ctypes.data_as(pd), d.ctypes.data_as(pd),
        p.ctypes.data_as(pd) if p is not None else None,
        C.c_uint32(self._flags(record_hits, histogram, record_segments))), 'odw_trace_rays')

  # -- batches: scenes of one structure in one launch -------------------------------
  def setSceneBatch(self, scenes):
    """scenes that differ in their numbers only (a parameter sweep: the same primitives, trimming lists, groups and
    optical types) side by side in HBM (`odw_upload_scene_batch`; limits first).  Scene 0 becomes the tracer's scene.
    Raises NativeError (unsupported) when the scenes differ in structure or lie outside the flat kernels' domain --
    the caller then traces them one by one."""
    if not len(scenes):
      raise ValueError('empty batch')
    descs, keeps = [], []
    for sc in scenes:
      if self.referenceStrict:
        import copy
        stripped = copy.copy(sc)
        stripped.prim_flags = np.asarray(sc.prim_flags, dtype=np.int32) & ~np.int32(_native.FLAG_CONVEX)
        sc = stripped
      if getattr(sc, 'surface_samplers', None):
        raise _native.NativeError('setSceneBatch: unsupported: scenes with stochastic surfaces are traced one by one')
      d, keep = _native.scene_desc(sc)
      descs.append(d)
      keeps.append(keep)
    arr = (_native.SceneDesc * len(descs))(*descs)
    self._chk(self._lib.odw_upload_scene_batch(self._ctx, arr, C.c_int32(len(descs))), 'odw_upload_scene_batch')
    arr0, n0, keep0 = _native.surface_sampler_descs(None)
    self._chk(self._lib.odw_upload_surface_samplers(self._ctx, arr0, C.c_int32(n0)), 'odw_upload_surface_samplers')
    self.scene = scenes[0]
    self.batchScenes = list(scenes)

  def reserveBatch(self, nScenes, rays, rowsPerScene):
    """room for batches of up to nScenes scenes (and for measuring them in HBM) now, not when the first large batch arrives"""
    self._chk(self._lib.odw_batch_reserve(self._ctx, C.c_int32(int(nScenes)), C.c_uint64(int(rays)), C.c_uint64(int(rowsPerScene))),
              'odw_batch_reserve')

  def traceBatch(self, first, n, seed, rowsPerScene, record_hits=True):
    """asynchronous: rays first..first+n-1 of Philox stream `seed` in EVERY scene of the batch, one launch; a scene's
    rows (those of trace() on that scene alone) go to its own segment of the batch's hit list"""
    self._chk(self._lib.odw_trace_batch(self._ctx, C.c_uint64(int(first)), C.c_uint64(int(n)), C.c_uint64(int(seed)),
                                        C.c_uint32(self._flags(record_hits, False)), C.c_uint64(int(rowsPerScene))),
              'odw_trace_batch')

  def batchSelect(self, k):
    """the rows of scene k of the last traceBatch become the tracer's hit list (hits(), hitCount(), deviceHits(),
    hitColumns() ...); None: back to the tracer's own list"""
    self._chk(self._lib.odw_batch_select(self._ctx, C.c_int32(-1 if k is None else int(k))), 'odw_batch_select')
    if k is not None:
      self.scene = self.batchScenes[int(k)]

  # -- a run's rows kept in HBM ----------------------------------------------------
  def archiveHits(self, source=None):
    """append the rows of `source`'s hit list (default: this tracer's own) to this tracer's archive in HBM
    (device to device; the source's launch is waited for) -> rows archived so far"""
    src = self if source is None else source
    n = C.c_uint64(0)
    f = self._lib.odw_archive_append
    f.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    self._chk(f(self._ctx, src._ctx, C.byref(n)), 'odw_archive_append')
    return int(n.value)

  def archiveSelect(self, on=True):
    """the archive becomes the tracer's hit list (hits(), hitCount(), deviceHits() ...); False: back to its own list"""
    self._chk(self._lib.odw_archive_select(self._ctx, C.c_int32(1 if on else 0)), 'odw_archive_select')

  def archiveReset(self):
    self._chk(self._lib.odw_archive_reset(self._ctx), 'odw_archive_reset')

  def batchRows(self):
    """(rows recorded per scene, slots asked for per scene) of the last traceBatch"""
    n = len(self.batchScenes)
    rows, wanted = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    pu = C.POINTER(C.c_uint64)
    self._chk(self._lib.odw_batch_rows(self._ctx, rows.ctypes.data_as(pu), wanted.ctypes.data_as(pu), C.c_int32(n)), 'odw_batch_rows')
    return rows, wanted

  def memInfo(self):
    """(free, total) bytes of the tracer's device"""
    free, total = C.c_uint64(0), C.c_uint64(0)
    self._chk(self._lib.odw_mem_info(self._ctx, C.byref(free), C.byref(total)), 'odw_mem_info')
    return int(free.value), int(total.value)

  def sync(self):
    self._chk(self._lib.odw_sync(self._ctx), 'odw_sync')

  def reset(self):
    self._chk(self._lib.odw_reset_results(self._ctx), 'odw_reset_results')

  def resetHits(self):
    """recycle the hit list (the reference's periodic flush,
    results_store.py:455-457); counters and histogram keep accumulating"""
    self._chk(self._lib.odw_reset_hits(self._ctx), 'odw_reset_hits')

  # -- results --------------------------------------------------------------
  def counters(self):
    out = np.zeros(len(CNT_NAMES), dtype=np.uint64)
    self._chk(self._lib.odw_fetch_counters(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                           C.c_int32(len(out))), 'odw_fetch_counters')
    return {k: int(v) for k, v in zip(CNT_NAMES, out)}

  @staticmethod
  def raiseForRayErrors(cnt):
    """exceptions the reference raises from inside Ray.traceRay, turned into counters by the kernels:
    a ray that enters a transmission grating while inside a medium (ray.py:234-237)"""
    if cnt.get('grating_in_medium'):
      raise ValueError('ray entered grating while already being inside a medium, get rid of any overlapping '
                       'lenses/transmission gratings in your project '
                       f'({cnt["grating_in_medium"]} ray(s) of this launch)')

  def hitCount(self):
    n = C.c_uint64(0)
    self._chk(self._lib.odw_hit_count(self._ctx, C.byref(n)), 'odw_hit_count')
    return int(n.value)

  def hits(self):
    """recorded hit rows sorted by (ray index, bounce order)"""
    n = self.hitCount()
    out = np.zeros(n, dtype=HIT_DTYPE)
    got = C.c_uint64(0)
    self._chk(self._lib.odw_fetch_hits(self._ctx, out.ctypes.data_as(C.c_void_p), C.c_uint64(n),
                                       C.byref(got)), 'odw_fetch_hits')
    return out[:int(got.value)]

  def hostRows(self, n):
    """page-locked array of n hit rows (destination of row fetches: copied into directly by the
    copy engine); freed with the tracer"""
    p = C.c_void_p()
    nbytes = int(n) * HIT_DTYPE.itemsize
    self._chk(self._lib.odw_host_alloc(self._ctx, C.c_uint64(nbytes), C.byref(p)), 'odw_host_alloc')
    self._pinned = getattr(self, '_pinned', [])
    self._pinned.append(p)
    buf = (C.c_char * nbytes).from_address(p.value)
    return np.frombuffer(buf, dtype=HIT_DTYPE)

  def traceStreaming(self, jobs, seed, capacity, histogram=True, buffers=None):
    """generator: trace every (first, n) of `jobs` and yield each job's hit rows on the host, the
    copy of job k overlapping the trace of job k+1 (two device hit lists, a copy stream of its
    own: odw_swap_hit_lists / odw_fetch_swapped_hits).  Rows come in append order (unordered
    across rays).  The arrays yielded are views of two alternating host buffers: use one before
    asking for the one after the next."""
    self.reserveHits(capacity)
    host = buffers or [self.hostRows(capacity) for _ in range(2)]
    got = C.c_uint64(0)

    def fetch(k):
      buf = host[k % 2]
      self._chk(self._lib.odw_fetch_swapped_hits(self._ctx, buf.ctypes.data_as(C.c_void_p), C.c_uint64(len(buf)),
                                                 C.byref(got)), 'odw_fetch_swapped_hits')
      return buf[:int(got.value)]
    # from the first swap on appends are dense (the rows of a list are copied as they lie)
    self._chk(self._lib.odw_swap_hit_lists(self._ctx), 'odw_swap_hit_lists')
    try:
      k = -1
      for k, (first, n) in enumerate(jobs):
        self.resetHits()
        self.trace(first, n, seed, histogram=histogram)       # asynchronous, into the current list
        if k > 0:
          yield fetch(k - 1)                                  # the list put aside: job k-1
        self._chk(self._lib.odw_swap_hit_lists(self._ctx), 'odw_swap_hit_lists')
      if k >= 0:
        yield fetch(k)
    finally:
      # back to one list: later launches reserve hit-list blocks per wave again
      self._chk(self._lib.odw_release_swapped_hits(self._ctx), 'odw_release_swapped_hits')

  def hitColumns(self, group, pinned=True, rayIndex=True):
    """the recorded rows of one group as the arrays the reference pickles per (source, object)
    (results_store.py:405-457): dict(points (n, 3), directions (n, 3), powers (n), isEntering (n) int64,
    rayIndex (n) int64), in (ray index, bounce) order -- selected and split into columns on the device
    (odw_hits_select + odw_hits_columns), so the host only receives them; None if the group has no row.
    pinned: the arrays live in page-locked memory of the process' pool (_PinnedPool) until they are dropped
    (they may outlive the tracer); then their memory returns to the pool.  rayIndex=False: without that column (it
    serves the per-ray metadata only: an eighth less to copy)"""
    n, leaving = C.c_uint64(0), C.c_uint64(0)
    self._chk(self._lib.odw_hits_select(self._ctx, C.c_int32(int(group)), C.byref(n), C.byref(leaving)), 'odw_hits_select')
    m = int(n.value)
    if m == 0:
      return None
    per_row = 9 if rayIndex else 8                          # doubles per row: 3 + 3 + 1 + 1 (+ 1: the ray's number)
    if pinned:
      buf = _POOL.take(self, m * 8 * per_row)
      f8 = np.frombuffer(buf, dtype=np.float64, count=per_row * m)
      out = dict(points=f8[:3 * m].reshape(m, 3), directions=f8[3 * m:6 * m].reshape(m, 3), powers=f8[6 * m:7 * m],
                 isEntering=f8[7 * m:8 * m].view(np.int64))
      if rayIndex:
        out['rayIndex'] = f8[8 * m:9 * m].view(np.int64)
    else:
      out = dict(points=np.empty((m, 3)), directions=np.empty((m, 3)), powers=np.empty(m), isEntering=np.empty(m, dtype=np.int64))
      if rayIndex:
        out['rayIndex'] = np.empty(m, dtype=np.int64)
    f = self._lib.odw_hits_columns
    f.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint64, C.POINTER(C.c_uint64)]
    got = C.c_uint64(0)
    self._chk(f(self._ctx, *(out[k].ctypes.data_as(C.c_void_p) if k in out else None
                             for k in ('points', 'directions', 'powers', 'isEntering', 'rayIndex')), C.c_uint64(m), C.byref(got)),
              'odw_hits_columns')
    assert int(got.value) == m
    return out

  def deviceHits(self, group=None):
    """the recorded rows as a `Hits`-like object that bins them where they are, in HBM
    (`simulation.device_hits.DeviceHits`); valid until the next launch, reset or fetch"""
    from .device_hits import DeviceHits
    self.sync()
    return DeviceHits(self, group)

  def loadHits(self, hits, group=0):
    """put recorded hits back into the device hit list (replacing its content) and return them as
    `DeviceHits`: `hits` = HIT_DTYPE rows, or a hit dictionary / `Hits` as `RawFolder.loadHits`
    returns it (points, directions, powers, isEntering; rows are numbered in their order)"""
    if isinstance(hits, np.ndarray) and hits.dtype == HIT_DTYPE:
      rows = np.ascontiguousarray(hits)
    else:
      d = getattr(hits, 'hits', hits)
      n = len(d['points'])
      rows = np.zeros(n, dtype=HIT_DTYPE)
      rows['point'], rows['direction'] = d['points'], d['directions']
      rows['power'] = d['powers'] if 'powers' in d else 1.0
      ent = np.asarray(d['isEntering'] if 'isEntering' in d else np.ones(n), dtype=np.uint64)
      rows['tag'] = (np.arange(n, dtype=np.uint64) | (np.uint64(int(group)) << np.uint64(48))
                     | ((ent != 0).astype(np.uint64) << np.uint64(63)))
    self._chk(self._lib.odw_load_hits(self._ctx, rows.ctypes.data_as(C.c_void_p), C.c_uint64(len(rows))), 'odw_load_hits')
    return self.deviceHits()

  def resetSegments(self):
    self._chk(self._lib.odw_reset_segments(self._ctx), 'odw_reset_segments')

  def segmentCount(self):
    """(rows in the segment list, rows that did not fit)"""
    n, dropped = C.c_uint64(0), C.c_uint64(0)
    self._chk(self._lib.odw_segment_count(self._ctx, C.byref(n), C.byref(dropped)), 'odw_segment_count')
    return int(n.value), int(dropped.value)

  def segments(self):
    """recorded ray segments sorted by (ray index, ordinal)"""
    n, _ = self.segmentCount()
    out = np.zeros(n, dtype=SEGMENT_DTYPE)
    got = C.c_uint64(0)
    self._chk(self._lib.odw_fetch_segments(self._ctx, out.ctypes.data_as(C.c_void_p), C.c_uint64(n),
                                           C.byref(got)), 'odw_fetch_segments')
    return out[:int(got.value)]

  def histogram(self):
    if self._det is None:
      raise ValueError('no detector set')
    nb = self._det['nx'] * self._det['ny']
    out = np.zeros(nb, dtype=np.uint64)
    self._chk(self._lib.odw_fetch_histogram(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            C.c_uint64(nb)), 'odw_fetch_histogram')
    return out.reshape(self._det['nx'], self._det['ny'])

  def sample(self, first, n, seed):
    t = np.empty(int(n))
    phi = np.empty(int(n))
    pd = C.POINTER(C.c_double)
    self._chk(self._lib.odw_sample(self._ctx, C.c_uint64(int(first)), C.c_uint64(int(n)),
                                   C.c_uint64(int(seed)), t.ctypes.data_as(pd), phi.ctypes.data_as(pd)),
              'odw_sample')
    return t, phi

  # -- device views for collectives ------------------------------------------
  def histogramView(self):
    p, n = C.c_void_p(), C.c_uint64(0)
    self._chk(self._lib.odw_device_histogram(self._ctx, C.byref(p), C.byref(n)), 'odw_device_histogram')
    return _CudaArrayView(p.value, n.value, '<i8', self)

  def countersView(self):
    p, n = C.c_void_p(), C.c_uint64(0)
    self._chk(self._lib.odw_device_counters(self._ctx, C.byref(p), C.byref(n)), 'odw_device_counters')
    return _CudaArrayView(p.value, n.value, '<i8', self)

  def resultsView(self):
    """counters and histogram as ONE int64 vector in HBM (`odw_device_results`): what a multi-GPU job sums with a
    single reduce.  -> (view, offset of the first histogram bin in words)"""
    p, n, off = C.c_void_p(), C.c_uint64(0), C.c_uint64(0)
    self._chk(self._lib.odw_device_results(self._ctx, C.byref(p), C.byref(n), C.byref(off)), 'odw_device_results')
    return _CudaArrayView(p.value, n.value, '<i8', self), int(off.value)

  def stream(self):
    p = C.c_void_p()
    self._chk(self._lib.odw_stream(self._ctx, C.byref(p)), 'odw_stream')
    return p.value

  # -- timing -----------------------------------------------------------------
  def timingEnable(self, on=True):
    self._chk(self._lib.odw_timing_enable(self._ctx, C.c_int(1 if on else 0)), 'odw_timing_enable')
    self._timingOn = bool(on)          # (contexts a sweep creates beside this one follow it: simulation/sweep.py)

  def timingRead(self):
    ms, n = C.c_double(0), C.c_uint64(0)
    self._chk(self._lib.odw_timing_read(self._ctx, C.byref(ms), C.byref(n)), 'odw_timing_read')
    return float(ms.value), int(n.value)


def segmentsToRays(segs, scene):
  """rows sorted by (ray, ordinal) -> one dictionary per ray, the layout
  SimulationResultsSingleRay.dump pickles (results_store.py:241-257):
  points (k+1, 3) = every segment's start + the end of the last one,
  powers (k,), media (k names, None = vacuum); plus globalRayIndex"""
  if len(segs) == 0:
    return []
  tags = segs['tag']
  ray = (tags & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
  medium = ((tags >> np.uint64(52)) & np.uint64(0xFFF)).astype(np.int64) - 1
  starts = np.flatnonzero(np.r_[True, ray[1:] != ray[:-1]])
  ends = np.r_[starts[1:], len(segs)]
  names = list(scene.group_names)
  out = []
  for a, b in zip(starts, ends):
    out.append(dict(points=np.vstack([segs['p1'][a:b], segs['p2'][b - 1:b]]),
                    powers=np.array(segs['power'][a:b]),
                    media=[names[m] if m >= 0 else None for m in medium[a:b]],
                    globalRayIndex=int(ray[a])))
  return out


def hitsToDict(hits, scene, sourceName='', group=None):
  """rows -> the reference's pickled hit dictionary
  (results_store.py:405-457): one dict per (source, object)."""
  tags = hits['tag']
  grp = ((tags >> np.uint64(48)) & np.uint64(0x7FFF)).astype(np.int64)
  out = {}
  for g in np.unique(grp) if group is None else [group]:
    sel = grp == g
    out[scene.group_names[g]] = dict(
        source=sourceName, obj=scene.group_names[g],
        points=np.ascontiguousarray(hits['point'][sel]),
        directions=np.ascontiguousarray(hits['direction'][sel]),
        powers=np.ascontiguousarray(hits['power'][sel]),
        isEntering=(tags[sel] >> np.uint64(63)).astype(np.int64),
        globalRayIndex=(tags[sel] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64))
  return out
