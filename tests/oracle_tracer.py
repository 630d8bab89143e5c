"""Test double of simulation.tracer.Tracer backed by the CPU oracle.

TEST INFRASTRUCTURE: lets the host logic above the C-ABI (runSimulation: modes,
end criteria, metadata, run folder, multi-rank sharding) run in the `not gpu`
suite.  Same method names and semantics as Tracer; results accumulate between
reset() calls like the device buffers do."""
import numpy as np

from oracle import capi

HIT_DTYPE = capi.HIT_DTYPE


class OracleTracer:

  def __init__(self, device=0, nthreads=4):
    self.device = device
    self._nthreads = nthreads
    self._det = None
    self._surface_seed = 0
    self._wavelength = None
    self.reset()

  # -- uploads ---------------------------------------------------------------
  def setScene(self, scene):
    self.scene = scene

  def setSource(self, source):
    self.source = source
    self._wavelength = None

  def setLimits(self, lim):
    self.limits = lim

  def setDetector(self, det):
    self._det = dict(det) if det is not None else None
    self._hist = None

  def setSurfaceSeed(self, seed):
    self._surface_seed = int(seed)

  def setWavelength(self, w):
    self._wavelength = float(w)

  def reserveHits(self, capacity):
    self._capacity = int(capacity)

  def reserveSegments(self, capacity):
    self._seg_capacity = int(capacity)

  # -- results ---------------------------------------------------------------
  def reset(self):
    self._block = None
    self._hits = []
    self._cnt = {k: 0 for k in capi.CNT_NAMES}
    self._hist = None
    self._segs = []

  def resetHits(self):
    self._hits = []

  def resetSegments(self):
    self._segs = []

  def _absorb(self, r):
    self._block = None
    for k, v in r['counters'].items():
      self._cnt[k] += v
    if len(r['hits']):
      self._hits.append(r['hits'])
    if 'hist' in r:
      self._hist = r['hist'] if self._hist is None else self._hist + r['hist']

  def trace(self, first, n, seed, record_hits=True, histogram=True, record_segments=False):
    if n == 0:
      return
    flags = (1 if record_hits else 0) | (2 if histogram else 0)
    if record_segments:
      if hasattr(self.source, 'face_prim'):
        o, d = capi.surface_rays(self.source, int(first), int(n), int(seed))
        g = capi.trace_segments(self.scene, self.limits, origins=o, dirs=d, wavelength=self.source.wavelength,
                                first=int(first), surface_seed=int(seed))
      else:
        g = capi.trace_segments(self.scene, self.limits, src=self.source, first=int(first), n=int(n), seed=int(seed))
      self._segs.append(g['segments'])
    if hasattr(self.source, 'face_prim'):
      r = capi.trace_surface(self.scene, self.source, self.limits, int(first), int(n), int(seed), det=self._det,
                             flags=flags, nthreads=self._nthreads)
    else:
      r = capi.trace(self.scene, self.source, self.limits, int(first), int(n), int(seed), det=self._det, flags=flags,
                     hit_capacity=int(n) * (self.limits.max_intersections + 1), nthreads=self._nthreads)
    self._absorb(r)

  def traceRays(self, origins, directions, powers=None, first=0, record_hits=True, histogram=True,
                record_segments=False):
    if self._wavelength is None:
      # the device keeps whatever odw_set_wavelength / odw_upload_source set last; a caller that
      # relies on that for explicit rays is a bug (every Ray carries its source's wavelength,
      # point_source.py:459) -- the double refuses instead of guessing
      raise RuntimeError('traceRays without setWavelength: explicit rays need their source\'s wavelength')
    wl = self._wavelength
    flags = (1 if record_hits else 0) | (2 if histogram else 0)
    if record_segments:
      self._segs.append(capi.trace_segments(self.scene, self.limits, origins=origins, dirs=directions, powers=powers,
                                            wavelength=wl, first=int(first),
                                            surface_seed=self._surface_seed)['segments'])
    self._absorb(capi.trace_rays(self.scene, self.limits, origins, directions, powers, wavelength=wl, first=int(first),
                                 det=self._det, flags=flags, nthreads=self._nthreads, surface_seed=self._surface_seed))

  def sync(self):
    pass

  RESULTS_HEAD = 16      # (the library's layout: counters, zero-padded to 16 words, then the histogram bins)

  def resultsView(self):
    """counters + histogram as one int64 block in host memory, the device library's layout (odw_device_results);
    what a reduce leaves in it is what counters() / histogram() report afterwards"""
    nb = self._det['nx'] * self._det['ny'] if self._det is not None else 0
    block = np.zeros(self.RESULTS_HEAD + nb, dtype=np.int64)
    block[:len(capi.CNT_NAMES)] = [self._cnt[k] for k in capi.CNT_NAMES]
    if nb:
      block[self.RESULTS_HEAD:] = self.histogram().astype(np.int64).ravel()
    self._block = block
    return block, self.RESULTS_HEAD

  def counters(self):
    if getattr(self, '_block', None) is not None:
      return {k: int(v) for k, v in zip(capi.CNT_NAMES, self._block)}
    return dict(self._cnt)

  def hitCount(self):
    return sum(len(h) for h in self._hits)

  def hits(self):
    if not self._hits:
      return np.zeros(0, dtype=HIT_DTYPE)
    h = np.concatenate(self._hits)
    ray = (h['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
    return h[np.argsort(ray, kind='stable')]

  def segmentCount(self):
    return sum(len(g) for g in self._segs), 0

  def segments(self):
    if not self._segs:
      return np.zeros(0, dtype=capi.SEGMENT_DTYPE)
    g = np.concatenate(self._segs)
    tag = g['tag']
    key = ((tag & np.uint64(0xFFFFFFFFFF)) << np.uint64(12)) | ((tag >> np.uint64(40)) & np.uint64(0xFFF))
    return g[np.argsort(key, kind='stable')]

  def histogram(self):
    if self._det is None:
      raise ValueError('no detector set')
    if getattr(self, '_block', None) is not None:
      return self._block[self.RESULTS_HEAD:].astype(np.uint64).reshape(self._det['nx'], self._det['ny'])
    if self._hist is None:
      return np.zeros((self._det['nx'], self._det['ny']), dtype=np.uint64)
    return self._hist

  def sample(self, first, n, seed):
    return capi.sample(self.source, int(first), int(n), int(seed))

  def generateRays(self, first, n, seed):
    if hasattr(self.source, 'face_prim'):
      return capi.surface_rays(self.source, int(first), int(n), int(seed))
    return capi.make_rays(self.source, int(first), int(n), int(seed))

  def close(self):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    self.close()
