"""Parameter sweeps: one Monte-Carlo run per parameter value, a figure of merit per run.

The reference's sweep is a notebook loop (examples/1-getting-started/
optimize-spotsize.ipynb cells 8-11; `ParameterSweeper`,
jupyter_utils/parameter_sweeper.py, wraps the same loop):

    for radius in radii:
      doc.Sphere.Radius = radius
      hits = doc.runSimulation('true').loadHits()
      fwhms.append(calcFwhm(hits))

Every run is independent, so the values are dealt out over the GPUs of the node
(one process per GPU, value k goes to rank k % world -- BASELINE configs[4]: 64
radii x 1e7 rays on 8 GPUs = 8 runs per GPU); the table value -> result is summed
into every rank with ONE collective at the end (each rank contributes its own
entries, zeros elsewhere).  Rays are addressed by the global Philox index, so a
value's result does not depend on which GPU ran it.
"""
import os
import sys
import time
import warnings

import numpy as np

from ..jupyter_utils.hits import Hits
from .. import _native
from ..scene import bake as _bake
from . import parallel
from .tracer import Tracer, hitsToDict

DEFAULT_SEED = 0x0D15EA5E


_FWHM_BINS = dict(binCoords='polar', bins=[np.arange(0, 2 * np.pi, np.pi / 2), np.geomspace(1e-3, 5, 500)])


def calcFwhm(hits):
  """spot FWHM of a hit cloud: `calcFwhm` of optimize-spotsize.ipynb cell 8, statement by
  statement -- polar histogram with bins [arange(0, 2 pi, pi/2), geomspace(1e-3, 5, 500)], per
  azimuth bin a straight-line fit of log(density) over log(r) through the first ten non-empty
  radial bins, FWHM = the smallest radius of the fitted line at or below half the peak density;
  mean over the azimuth bins that received hits (nan if none did)"""
  return _fwhmOfPolarHistogram(hits.histogram(**_FWHM_BINS))


def _fwhmOfPolarHistogram(polarHist):
  """cell 8 from `phis, r, hists = polarHist.byAzimuth()` on"""
  fwhmList = []
  phis, r, hists = polarHist.byAzimuth()
  for phi, dens in zip(phis, hists):
    if max(dens) > 0:
      with warnings.catch_warnings():
        # (numpy's RankWarning where fewer than two radial bins hold hits: the notebook prints it, a sweep of
        #  64 values would print it hundreds of times; the fit's numbers are the same either way)
        warnings.simplefilter('ignore')
        a, b = np.polyfit(np.log(r[dens > 0])[:10], np.log(dens[dens > 0])[:10], deg=1)
      # (outside the try, as in the notebook: an azimuth bin without a radial bin above 10 -- max() of an
      #  empty selection -- raises ValueError to the caller; with the notebook's own 1e3 rays that happens)
      rFit = np.geomspace(min(r), max(r[dens > 10][:10]), 100)
      fitDens = np.exp(a * np.log(rFit) + b)
      try:
        fwhmList.append(min(rFit[fitDens <= max(dens) / 2]))
      # the notebook ignores the cases where the FWHM is ill defined (the fit never falls to half the peak)
      except ValueError:
        pass
  return np.mean(fwhmList) if len(fwhmList) else np.nan


# a measure may carry `batched(DeviceHitsBatch) -> list`: its values for all scenes of a batch launch at once (None
# where a scene has to be measured by itself) -- parameterSweep then waits for the GPU once per step and batch
# instead of once per step and value
calcFwhm.batched = lambda batch: [None if H is None else _fwhmOfPolarHistogram(H) for H in batch.histograms(**_FWHM_BINS)]
calcFwhm.batchedBins = _FWHM_BINS        # (the histogram `batched` asks for: a chain bins it on its way)


def rmsSpot(hits):
  """rms distance of the hits from their centroid: a figure of merit that stays defined where the
  notebook's calcFwhm is not (with 1e7 hits the innermost ten radial bins are all filled and flat:
  the fitted line never falls to half the peak, the notebook skips such azimuth bins and returns
  nan near the focus)"""
  if hasattr(hits, 'rmsSpot'):
    return hits.rmsSpot()                 # DeviceHits: second moments on the device
  p = hits.points()
  return float(np.sqrt(((p - p.mean(axis=0))**2).sum(axis=1).mean())) if len(p) else np.nan


rmsSpot.batched = lambda batch: [None if m is None else float(np.sqrt(m[1].sum())) for m in batch.moments()]


def _rowsToDict(rows):
  return dict(points=np.ascontiguousarray(rows['point']), directions=np.ascontiguousarray(rows['direction']),
              powers=np.ascontiguousarray(rows['power']), isEntering=(rows['tag'] >> np.uint64(63)).astype(np.int64))


def _thinnedRows(hits, count):
  """`[::max(1, n // count)]` of a run's hit arrays (in the order `loadHits()` returns them: by ray, then bounce)"""
  if hasattr(hits, 'thinned'):
    return _rowsToDict(hits.thinned(count))              # DeviceHits: gathered on the device
  k = max(1, len(hits) // int(count))
  return dict(points=hits.points()[::k].copy(), directions=hits.directions()[::k].copy(),
              powers=np.asarray(hits.hits['powers'])[::k].copy(), isEntering=np.asarray(hits.isEntering())[::k].copy())


def fwhmOfSamples(result, dist=None, device=0, measure=None):
  """`calcFwhm` (or `measure`) of the thinned samples a sweep kept (`keepSample`): the notebook's figure of merit on
  the notebook's own sample size.  Every rank evaluates the samples of the values it ran; one all-reduce gives every
  rank the whole column (nan where a value has no sample or the notebook's arithmetic raises)."""
  measure = measure or calcFwhm
  ranks = parallel.Ranks.detect(dist, device)
  col = np.zeros((len(result.values), 2))
  for k, d in result.samples.items():
    try:
      m = float(measure(Hits(d)))
    except ValueError:                 # (cell 8 raises where an azimuth bin holds no radial bin above 10 counts)
      m = np.nan
    col[k] = (0.0, 2.0) if np.isnan(m) else (m, 1.0)
  col = np.asarray(ranks.sumFloats(col.ravel())).reshape(col.shape)
  return np.where(col[:, 1] == 1, col[:, 0], np.nan)


def shareOfRank(n_values, rank, world):
  """indices of the sweep values rank `rank` runs: k = rank, rank + world, ..."""
  return list(range(int(rank), int(n_values), int(world)))


def _sourceKey(bsrc):
  """what decides a source's device tables: the (cached) sampler tables by identity, the numbers by value"""
  t = getattr(bsrc, 'tables', None)
  if t is None:
    return id(bsrc)
  return (id(t), np.asarray(bsrc.xform, dtype=np.float64).tobytes(), float(bsrc.focal_length), float(bsrc.wavelength),
          float(bsrc.power))


class SweepResult:
  """values, one column of results per figure of merit (nan where a run produced none), and what
  the runs traced.  `results` is the first (or only) column."""

  def __init__(self, values, columns, tracedRays, recordedHits, segments, samples=None):
    self.samples = samples or {}      # value index -> thinned hit rows of this rank's own values (keepSample)
    self.values = np.asarray(values, dtype=np.float64)
    self.columns = {k: np.asarray(v, dtype=np.float64) for k, v in columns.items()}
    self.results = next(iter(self.columns.values()))
    self.tracedRays, self.recordedHits, self.segments = int(tracedRays), int(recordedHits), int(segments)

  def best(self, column=None):
    """(value, result) of the smallest finite result (the notebook's `radii[np.argmin(fwhms)]`)"""
    col = self.results if column is None else self.columns[column]
    k = int(np.nanargmin(col))
    return float(self.values[k]), float(col[k])


def parameterSweep(doc, setValue, values, *, rays, measure=calcFwhm, seed=DEFAULT_SEED, device=0,
                   dist=None, tracer=None, source=None, deviceHits=True, pipeline=True, batch=12, keepSample=None,
                   **traceKwargs):
  """run `rays` true-random rays for every entry of `values` and return a SweepResult.

  setValue(doc, value)   applies one parameter value (e.g. `doc.Sphere.Radius = value`)
  measure                figure of merit of a run's hit cloud: callable(hits) -> float, or a dict
                         name -> callable for several columns (`Hits` interface: histogram, points,
                         directions ...); default: the notebook's calcFwhm
  dist                   torch.distributed (initialised): the values are dealt out over the ranks
                         and the table is summed into every rank with one all-reduce
  deviceHits             measure on the hit rows where they are, in HBM (`DeviceHits`: plane search
                         on a thinned sample, projection, medians and binning on the device); False:
                         copy every row to the host first (what the reference does)
  batch                  with deviceHits on a device tracer: so many values at a time are traced by ONE launch (scenes of
                         one structure side by side in HBM, `Tracer.traceBatch`; the rows of a value are those of a launch of
                         its own); values whose scenes differ in structure, or that the flat kernels do not take, are traced
                         one by one; 0 / 1: always one by one
  keepSample             an integer N: the rows `[::max(1, n // N)]` of every value's hit list (the sample a notebook that
                         traces N rays per value works on) are kept in `SweepResult.samples` on the rank that ran the value
                         (`fwhmOfSamples` turns them into a column)
  pipeline               with deviceHits on a device tracer: further contexts on the same GPU (True: three), so
                         that the next group of values is baked and traced while earlier ones are measured (same
                         results); an integer n: n extra contexts (n measuring threads)
  """
  # (collectives run on the GPU the tracer works on: one process per GPU, each with its own device)
  ranks = parallel.Ranks.detect(dist, getattr(tracer, 'device', device) if tracer is not None else device)
  values = [float(v) for v in values]
  measures = dict(measure) if isinstance(measure, dict) else {'result': measure}
  names = list(measures)
  mine = shareOfRank(len(values), ranks.rank, ranks.world)
  sources = _bake.lightSources(doc)
  if not sources:
    raise ValueError('document has no light source')
  src = sources[0] if source is None else source
  from .results_store import updateResultEntry
  from .simulation_loop import bakeLightSource
  own = tracer is None
  tr = tracer or Tracer(device)
  # Contexts on the same GPU take turns: while worker threads measure the rows of earlier values on theirs
  # (selection, the plane search on the host, projection, binning -- GPU work and host work in alternation),
  # the main thread bakes and traces the next value on a free one.  Each context has its own stream and hit list;
  # the compiled kernel is shared through the process cache.  Only for device tracers measured in HBM.
  lanes = [tr]
  if os.environ.get('ODW_SWEEP_PIPELINE') == '0':        # (diagnostics: every kernel of a sweep by itself in a trace)
    pipeline = False
  elif os.environ.get('ODW_SWEEP_PIPELINE'):
    pipeline = int(os.environ['ODW_SWEEP_PIPELINE'])
  if pipeline and deviceHits and isinstance(tr, Tracer) and len(mine) > 1:
    # (three extra contexts by default: groups of values take turns on four contexts -- while the chain of one waits for its
    #  plane search or its fits, the launches and chains of the others keep the GPU busy.  64 x 1e7 rays, batch 12, round 5:
    #  1 / 2 / 3 / 4 extra contexts = 90 / 83 / 74 / 79 ms per sweep.  Each context holds a group's hit list and the chain's
    #  buffers: ~17 GB at 12 values x 1.25e7 rows.  The extra contexts stay with the tracer between sweeps)
    want = min(int(pipeline) if pipeline is not True else 3, len(mine) - 1, 5)
    kept = [e for e in (getattr(tr, '_sweepLanes', None) or [])
            if e.referenceStrict == tr.referenceStrict and e.compileMode() == tr.compileMode()]
    while len(kept) < want:
      extra = Tracer(tr.device, referenceStrict=tr.referenceStrict)
      try:
        extra.compileScene({0: 'off', 1: 'structure', 2: 'auto'}[tr.compileMode()])
      except Exception:
        pass
      if getattr(tr, '_timingOn', False):
        extra.timingEnable(True)         # (a tracer whose launches are being timed: those of its new contexts too)
      kept.append(extra)
    tr._sweepLanes = kept
    lanes += kept[:want]
  uploaded = [dict() for _ in lanes]       # (what each context holds is looked at afresh every sweep)
  table = np.zeros((len(values), len(names), 2))        # (result or 0, 1 = a number / 2 = nan)
  totals = np.zeros(3, dtype=np.int64)
  pool = None
  pending = [None] * len(lanes)
  batch_ok = [True]
  samples = {}
  tail_size = [None]
  taper = int(os.environ.get('ODW_SWEEP_TAPER', '0'))

  def measureInto(t, scene, k):
    t_m = time.perf_counter() if clock is not None else 0.0
    try:
      _measure(t, scene, k)
    finally:
      if clock is not None:
        clock['measure'] += time.perf_counter() - t_m      # (summed over the measuring threads)

  def _measure(t, scene, k):
    if deviceHits and hasattr(t, 'deviceHits'):
      hits = t.deviceHits()
    else:
      merged = {}
      for d in hitsToDict(t.hits(), scene, src.Name).values():
        for key, v in d.items():
          updateResultEntry(merged, key, v)
      hits = Hits(merged)
    for j, name in enumerate(names):
      m = float(measures[name](hits)) if len(hits) else np.nan
      table[k, j] = (0.0, 2.0) if np.isnan(m) else (m, 1.0)
    if keepSample and len(hits):
      samples[k] = _thinnedRows(hits, int(keepSample))

  clock = dict(wait=0.0, bake=0.0, trace=0.0, measure=0.0) if os.environ.get('ODW_SWEEP_TIMING') else None
  import threading
  timeline = [] if os.environ.get('ODW_SWEEP_TRACE') else None
  t_sweep = time.perf_counter()

  def mark(what, t0):
    if timeline is not None:
      timeline.append((threading.current_thread().name[-12:], what, 1e3 * (t0 - t_sweep), 1e3 * (time.perf_counter() - t_sweep)))
  totals_lock = threading.Lock()

  def bakeValue(k):
    setValue(doc, values[k])
    return _bake.bakeScene(doc, src), bakeLightSource(doc, src, seed), _bake.bakeLimits(doc, src, **traceKwargs)

  def uploadCommon(t, up, bsrc, lim):
    # (tables travel to the device only when they change: a sweep of one shape parameter uploads
    #  the source's 1.6 MB of sampler tables once)
    key = _sourceKey(bsrc)
    if up.get('source') != key:
      t.setSource(bsrc)
      up['source'] = key
    if up.get('limits') != lim:
      t.setLimits(lim)
      up['limits'] = lim

  def runOne(lane, k, baked=None):
    """one value, one launch (tracers without batch launches, scenes outside the flat kernels' domain, values whose
    scenes differ in structure)"""
    t, up = lanes[lane], uploaded[lane]
    t0 = time.perf_counter()
    if pending[lane] is not None:
      pending[lane].result()          # the rows of this context are free again (and its errors surface here)
      pending[lane] = None
    t1 = time.perf_counter()
    scene, bsrc, lim = baked if baked is not None else bakeValue(k)
    t2 = time.perf_counter()
    if hasattr(t, 'batchSelect') and getattr(t, 'batchScenes', None):
      t.batchSelect(None)
    uploadCommon(t, up, bsrc, lim)
    t.setScene(scene)
    t.setDetector(None)
    capacity = int(rays * 1.25) + 1024
    while True:
      t.reserveHits(capacity)
      t.reset()
      t.trace(0, int(rays), seed, histogram=False)
      t.sync()
      cnt = t.counters()
      Tracer.raiseForRayErrors(cnt)
      if not cnt['hits_dropped']:
        break
      capacity = int(cnt['recorded_hits'] * 1.05) + 1024      # deterministic: trace again with room
    with totals_lock:
      totals[:] += (cnt['traced_rays'], cnt['recorded_hits'], cnt['segments'])
    if clock is not None:
      t3 = time.perf_counter()
      clock['wait'] += t1 - t0; clock['bake'] += t2 - t1; clock['trace'] += t3 - t2
    if pool is not None:
      pending[lane] = pool.submit(measureInto, t, scene, k)
    else:
      measureInto(t, scene, k)

  def histogramRequest():
    """the histogram the measures' batched forms will ask a DeviceHitsBatch for (the chain bins it on the way), or None"""
    for name in names:
      req = getattr(measures[name], 'batchedBins', None)
      if req is not None and hasattr(measures[name], 'batched'):
        return req
    return None

  def finishGroup(t, ks, batch_hits):
    """the measures of a group whose chain has arrived (DeviceHitsBatch in state 'measured'), scene by scene"""
    t_m = time.perf_counter()
    try:
      together = {}
      for name in names:
        if hasattr(measures[name], 'batched'):
          together[name] = measures[name].batched(batch_hits)
      kept = batch_hits.thinned(int(keepSample)) if keepSample else None
      for j, k in enumerate(ks):
        own_hits = None
        if keepSample:
          if kept is not None and kept[j] is not None:
            if len(kept[j]):
              samples[k] = _rowsToDict(kept[j])
          else:
            t.batchSelect(j)
            own_hits = t.deviceHits()
            if len(own_hits):
              samples[k] = _thinnedRows(own_hits, int(keepSample))
        for i, name in enumerate(names):
          m = together[name][j] if name in together else None
          if batch_hits.rows[j] == 0:
            m = np.nan
          elif m is None:                     # this measure, or this scene, goes segment by segment
            if own_hits is None:
              t.batchSelect(j)
              own_hits = t.deviceHits()
            m = float(measures[name](own_hits)) if len(own_hits) else np.nan
          m = float(m)
          table[k, i] = (0.0, 2.0) if np.isnan(m) else (m, 1.0)
    finally:
      if t is not None:
        t.batchSelect(None)
      mark(f'finish {ks[0]}', t_m)
      if clock is not None:
        clock['measure'] += time.perf_counter() - t_m

  def sweepBatched(group_size):
    """Batch launches driven as chains from THIS thread: a group of values is baked, traced by one launch and its
    measure enqueued behind it on a context of its own (`DeviceHitsBatch.begin`); the thread then turns to whatever is
    ready -- samples that have arrived (their plane searches, in lockstep for all groups that are ready, then the rest of
    the chain is enqueued), chains that have finished (fits, table) -- and launches the next group when a context is free.
    No worker threads: nothing here waits for the GPU while there is host work to do, and the host work (bake, plane
    search, fits) never competes for the interpreter lock."""
    from .device_hits import DeviceHitsBatch
    request = histogramRequest()
    while True:
      try:
        for t in lanes:
          # (every context sized for the largest group before the first launch: a hit list that has to grow later is
          #  released and allocated again, which waits for every stream of the device -- 13 GB take half a second)
          key = (group_size, int(rays))
          if getattr(t, '_sweepReserved', None) != key:
            t.reserveBatch(group_size, int(rays), int(rays * 1.25) + 1024)
            t._sweepReserved = key
            t._sweepHeld = group_size * ((int(rays * 1.25) + 1024 + 4_300_000) * (64 + 24 + 4) + int(rays) * 30)
        break
      except _native.NativeError as e:
        # the device has less room than the estimate said (someone else's buffers): half the group, down to values one by one
        if 'device error' not in str(e) or group_size <= 1:
          raise
        group_size = max(1, group_size // 2)
        for t in lanes:
          t._sweepReserved = None
        if group_size == 1:
          for turn_, k in enumerate(mine):
            runOne(turn_ % len(lanes), k)
          return
    busy = [None] * len(lanes)           # per context: dict(ks, batch, capacity, t0) of the group in flight
    order = []                           # contexts in the order their groups were launched
    pos, turn = 0, 0
    # host work nobody waits for is done when the GPU has been fed: the fits of a group whose answers are all on the host
    # (its context is free at once), and the bake of the group that goes next (ODW_SWEEP_DEFER=0: both where they used to be)
    defer = os.environ.get('ODW_SWEEP_DEFER', '1') != '0' and all(hasattr(measures[name], 'batched') for name in names)
    deferred = []                        # (ks, batch) whose fits are still to do
    ahead = [None]                       # (ks, baked) of the next group, baked in an idle moment

    def launch(lane, ks, baked):
      t, up = lanes[lane], uploaded[lane]
      uploadCommon(t, up, baked[0][1], baked[0][2])
      t.setSceneBatch([b[0] for b in baked])
      t.setDetector(None)
      capacity = int(rays * 1.25) + 1024
      t.reset()
      t.traceBatch(0, int(rays), seed, capacity)
      busy[lane] = dict(ks=ks, batch=DeviceHitsBatch.begin(t, len(ks)), capacity=capacity)
      order.append(lane)

    def sampledGroups(wait_lane=None):
      """contexts whose sample has arrived: counters checked (a segment without room: traced again), planes searched
      together, the rest of the chain enqueued"""
      ready = []
      for lane in list(order):
        g = busy[lane]
        if g['batch']._stage == 'begun' and g['batch'].sampled(wait=(lane == wait_lane)):
          t = lanes[lane]
          cnt = t.counters()
          Tracer.raiseForRayErrors(cnt)
          if cnt['hits_dropped']:
            rows, wanted = t.batchRows()                              # deterministic: trace again with room
            g['capacity'] = int(int(wanted.max()) * 1.05) + 1024
            t.reset()
            t.traceBatch(0, int(rays), seed, g['capacity'])
            g['batch'] = DeviceHitsBatch.begin(t, len(g['ks']))
            continue
          totals[:] += (cnt['traced_rays'], cnt['recorded_hits'], cnt['segments'])
          ready.append(lane)
      if ready:
        t_s = time.perf_counter()
        DeviceHitsBatch.searchPlanesTogether([busy[lane]['batch'] for lane in ready])
        mark(f'planes {[busy[lane]["ks"][0] for lane in ready]}', t_s)
        for lane in ready:
          busy[lane]['batch'].enqueueMeasure(keep=int(keepSample or 0), **(request or {}))
      return bool(ready)

    def measuredGroups(wait_lane=None):
      done = False
      for lane in list(order):
        g = busy[lane]
        if g['batch']._stage == 'measuring' and g['batch'].measured(wait=(lane == wait_lane)):
          if defer and g['batch'].detached():
            deferred.append((g['ks'], g['batch']))
          else:
            finishGroup(lanes[lane], g['ks'], g['batch'])
          busy[lane] = None
          order.remove(lane)
          done = True
      return done

    def nextGroup():
      """the values of the next group, baked (the sizes: see below)"""
      nonlocal turn
      if ahead[0] is not None:
        out, ahead[0] = ahead[0], None
        return out
      # (the first groups are small, so that chains start early; the last ones shrink, so that the contexts end together.
      #  ODW_SWEEP_TAPER = d > 0: every group of the tail takes 1 / d of what is left; 0: the tail in equal groups -- measured
      #  better: small launches are worse launches)
      left = len(mine) - pos
      if group_size > 2 and left < group_size * len(lanes):
        if taper > 0:
          tail_size[0] = max(2, -(-left // taper))
        elif tail_size[0] is None:
          tail_size[0] = max(2, -(-left // len(lanes)))
      size = group_size if group_size <= 2 else min(group_size, 2 << turn if turn < 3 else group_size, tail_size[0] or group_size)
      ks = mine[pos:pos + size]
      turn += 1
      t1 = time.perf_counter()
      baked = [bakeValue(k) for k in ks]
      mark(f'bake {ks[0]}', t1)
      return ks, baked

    while pos < len(mine) or order or deferred:
      did = measuredGroups()
      did = sampledGroups() or did
      free = [lane for lane in range(len(lanes)) if busy[lane] is None]
      if pos < len(mine) and free and batch_ok[0]:
        ks, baked = nextGroup()
        same = all(b[2] == baked[0][2] and _sourceKey(b[1]) == _sourceKey(baked[0][1]) for b in baked[1:])
        launched = False
        if same and len(ks) > 1:
          t2 = time.perf_counter()
          try:
            launch(free[0], ks, baked)
            launched = True
            mark(f'upload+launch {ks[0]}', t2)
          except _native.NativeError as e:
            if 'unsupported' not in str(e):
              raise
            batch_ok[0] = False           # (another structure per value, or scenes the flat kernels do not take: one by one)
        if not launched:
          for k, b in zip(ks, baked):
            runOne(free[0], k, baked=b)
        pos += len(ks)
        continue
      if pos < len(mine) and not batch_ok[0] and not order:
        for k in mine[pos:]:
          runOne(0, k)
        pos = len(mine)
        continue
      if not did and defer and batch_ok[0] and pos < len(mine) and ahead[0] is None and order:
        ahead[0] = nextGroup()             # every context is busy: the next group's bake, so that its launch is an upload away
        continue
      if not did and deferred:
        ks_d, batch_d = deferred.pop(0)    # nothing to feed the GPU with: the fits of a group that has arrived
        finishGroup(None, ks_d, batch_d)
        continue
      if not did and order:
        # nothing ready and nothing to launch: until ANY chain's next piece has arrived.  (Waiting for the oldest chain's
        # piece blocks the thread while a younger chain's sample lies ready -- its plane search, its measure and the launch
        # that needs its context all start late: sweeps of 65 instead of 59 ms, every other one, measured.)
        t_w = time.perf_counter()
        if os.environ.get('ODW_SWEEP_WAIT_OLDEST') == '1':
          lane = order[0]
          if busy[lane]['batch']._stage == 'begun':
            sampledGroups(wait_lane=lane)
          else:
            measuredGroups(wait_lane=lane)
        else:
          while not (measuredGroups() or sampledGroups()):
            time.sleep(2e-5)
        mark('wait', t_w)

  # Batch launches (Tracer.setSceneBatch / traceBatch): the values a context gets at a time are baked together and traced
  # by ONE launch -- their scenes differ in numbers only --, each into its own segment of the hit list; a measuring thread
  # then goes through the segments while the main thread bakes and launches the next group on another context.
  group_size = int(batch) if (batch and deviceHits and isinstance(tr, Tracer) and len(mine) > 1) else 1
  if group_size > 1:
    # room in HBM: every context holds the hit list of a group (64-byte rows, with the slack of block reservations) and
    # the post-hoc state of its segments (row-of-ray table, selection, projected coordinates: ~30 bytes per ray)
    per_value = (int(rays * 1.25) + 1024 + 4_300_000) * (64 + 24 + 4) + int(rays) * 30
    # (what the device has free NOW, shared with whatever else lives on it -- other ranks of a rehearsal, the caller's own
    #  buffers --, never more than ODW_SWEEP_HBM_GB; two thirds of it, over this sweep's contexts)
    budget = float(os.environ.get('ODW_SWEEP_HBM_GB', '128')) * 1e9
    try:
      free, _ = tr.memInfo()
      held = sum(getattr(t, '_sweepHeld', 0) for t in lanes)        # (what these contexts reserved in earlier sweeps is theirs)
      budget = min(budget, (free + held) * 2 / 3)
    except Exception:
      pass
    # (ranks that share one device -- a node rehearsed on one GPU -- share its memory too, and ask at the same moment)
    budget /= max(1, int(os.environ.get('ODW_RANKS_PER_DEVICE', '1')))
    group_size = max(1, min(group_size, int(budget / len(lanes) // per_value)))
  if os.environ.get('ODW_SWEEP_BATCH'):
    group_size = max(1, int(os.environ['ODW_SWEEP_BATCH'])) if group_size > 1 else 1
  switch_interval = sys.getswitchinterval()
  if group_size <= 1 and len(lanes) > 1:
    # values one by one: measuring threads, one per context (batch launches are driven as chains from this thread alone)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=len(lanes), thread_name_prefix='odw-sweep-measure')
  if pool is not None:
    # (threads that alternate between short library calls and a few lines of Python hand the interpreter lock to
    #  each other all the time; with the default 5 ms a thread that comes back from a 20 us call can wait that long)
    sys.setswitchinterval(float(os.environ.get('ODW_SWITCH_INTERVAL', '2e-4')))
  try:
    if group_size > 1:
      sweepBatched(group_size)
    else:
      for turn, k in enumerate(mine):
        runOne(turn % len(lanes), k)
    for f in pending:
      if f is not None:
        f.result()
  finally:
    sys.setswitchinterval(switch_interval)
    if pool is not None:
      pool.shutdown(wait=True)
    if own:
      tr.close()                       # (with its extra contexts)
  if timeline is not None:
    for row in sorted(timeline, key=lambda r: r[2]):
      print('[odw sweep trace] %-12s %-22s %8.2f -> %8.2f  (%6.2f)' % (row[0], row[1], row[2], row[3], row[3] - row[2]), file=sys.stderr)
    print('[odw sweep trace] total %.2f ms' % (1e3 * (time.perf_counter() - t_sweep)), file=sys.stderr, flush=True)
  if clock is not None:
    print('[odw sweep timing] ms per value: ' + ', '.join(f'{k} {1e3 * v / max(len(mine), 1):.2f}' for k, v in clock.items()),
          file=sys.stderr, flush=True)
  flat = ranks.sumFloats(np.concatenate([table.ravel(), totals.astype(np.float64)]))
  table = np.asarray(flat[:table.size]).reshape(table.shape)
  if not np.all((table[..., 1] == 1) | (table[..., 1] == 2)):
    raise RuntimeError('parameter sweep: some values were run by no rank or by several')
  results = np.where(table[..., 1] == 1, table[..., 0], np.nan)
  t = [int(round(v)) for v in flat[table.size:]]
  return SweepResult(values, {name: results[:, j] for j, name in enumerate(names)}, *t, samples=samples)
