"""runSimulation: the continuous Monte-Carlo run on the GPU tracer.

Counterpart of `simulation.runSimulation(action)` main loop A
(simulation/processes/simulation_loop.py:291-632): for every light source an
iteration generates RaysPerIteration*RaysPerIterationScale rays and traces
them (generic_source.py:51-146); the loop ends when totalIterations >
EndAfterIterations, totalTracedRays > EndAfterRays or totalRecordedHits >
EndAfterHits (results_store.py:486-512).  Here many iterations are fused into
one device launch; the end criteria are evaluated between launches, so the
overshoot is at most one launch (the reference overshoots too: its criteria
are evaluated asynchronously from worker progress files).

Worker processes, flag files and the FreeCAD child process of the reference
are replaced by the device; `devices` > 1 shards the ray index range over
several GPUs of this process' node through `parallel`.
"""
import os
import threading
import time

import numpy as np

from ..freecad_elements import point_source, replay_source, surface_fans, surface_source
from ..scene import bake as _bake
from . import parallel, results_store
from .tracer import Tracer, segmentsToRays

DEFAULT_SEED = 0x0D15EA5E
_RECORD_RAYS_PER_LAUNCH = 1 << 16
_MODES = ('true', 'singletrue', 'pseudo', 'singlepseudo', 'fans', 'singlefans')


def _limit(settings, key, default):
  v = settings._props.get(key, default) if settings is not None else default
  try:
    return float(v)
  except (TypeError, ValueError):
    return np.inf


def runSimulation(doc, action='true', *, seed=DEFAULT_SEED, device=0, resultsPath=None, store=None,
                  raysPerLaunch=1 << 22, endIf=None, tracer=None, pseudoIterationsPerLaunch=64,
                  dist=None, compileScene='auto', overlapFetch=True, keepOnDevice='auto', **traceKwargs):
  """trace `doc` until its simulation settings' end criteria are met.

  action       'true' (continuous Monte-Carlo) | 'singletrue' (one iteration)
               | 'pseudo' / 'singlepseudo' (histogram-thinned draws of
               VectorRandomVariable.drawPseudo, one call per iteration as in
               point_source.py:670; the draws are the reference's serial host
               algorithm, the rays are traced on the device)
               | 'fans' / 'singlefans' (ray fans, explicit initial conditions)
  resultsPath  `<doc>.OpticsDesign` folder to write the run folder into
               (None: keep results in memory only)
  endIf        optional callback(store) -> bool, checked between launches
               (FreecadDocument.runSimulation's endIf)
  dist         torch.distributed (initialised): one process per GPU, every launch's
               ray index range is sharded over the ranks, each rank writes its own
               hit files into the shared run folder, the end criteria see the
               job's totals (`parallel.Ranks`); picked up automatically under a
               launcher (WORLD_SIZE > 1)
  compileScene 'auto': a continuous run lets the library compile the ray loop against the scene in the
               background once the scene has traced 5e7 rays, and switches over when the kernel is ready
               (Tracer.compileScene('auto'): nothing waits, the results are those of the generic kernels
               bit for bit); 'structure': compile before the first launch; 'off': never.  Applies to
               the tracer this call creates; a tracer passed in keeps its own setting.
  overlapFetch a continuous true-random run of ONE point or surface source whose end criteria do not look at the hits
               (EndAfterHits inf, no endIf), in one process: launches alternate between two contexts of the GPU, and
               while one traces, the hit columns of the other's previous launch cross PCIe into page-locked arrays and
               go to the writer threads (a thread of this call does that) -- the same files, the order of rays in
               them included; False: one launch at a time, fetched before the next starts
  keepOnDevice the run's hit rows also stay in HBM, appended launch by launch to an archive on the device (a context of its
               own): `store.deviceHits()` -- and `RawFolder.loadHits()` of the same process -- then bin them where they
               are (`DeviceHits`) instead of reading the run folder back into host arrays (freecad_document.py:1485-1504);
               the files are written all the same.  'auto' (default) / True: up to 64 GB of rows and at most a third of
               the device memory that is free when the run starts; a number: that many GB; False: never.  Applies to one
               process tracing one source; a run that outgrows its budget drops the archive and goes on (the run folder
               is then the only copy).
  traceKwargs  maxRayLength, maxIntersections, powerTol, distTol (ray.py:36-38)
  -> SimulationResults
  """
  if action not in _MODES:
    raise ValueError(f'unexpected simulation mode {action!r}')
  settings = _bake.activeSimulationSettings(doc)
  continuous = action in ('true', 'pseudo')
  pseudo = action in ('pseudo', 'singlepseudo')
  if pseudo:
    # the reference seeds numpy's global generator per worker (simulation_loop.py:813-820)
    np.random.seed(int(seed) % (1 << 32))
  # (collectives run on the GPU the tracer works on: one process per GPU, each with its own device)
  ranks = parallel.Ranks.detect(dist, getattr(tracer, 'device', device) if tracer is not None else device)
  if store is None:
    runFolder = None
    if resultsPath is not None and ranks.world > 1:
      # every rank writes into the run folder rank 0 creates
      runFolder = ranks.broadcast(f'raw/simulation-run-{results_store.latestRunIndex(resultsPath) + 1:06d}'
                                  if ranks.rank == 0 else None)
    store = results_store.SimulationResults(
        action, resultsPath=resultsPath, simulationRunFolder=runFolder,
        endAfterIterations=_limit(settings, 'EndAfterIterations', np.inf) if continuous else 0,
        endAfterRays=_limit(settings, 'EndAfterRays', np.inf) if continuous else np.inf,
        endAfterHits=_limit(settings, 'EndAfterHits', np.inf) if continuous else np.inf,
        owner=ranks.rank == 0)
    ranks.barrier()
  if continuous and not (np.isfinite(store.endAfterIterations) or np.isfinite(store.endAfterRays)
                         or np.isfinite(store.endAfterHits)) and endIf is None:
    raise ValueError('continuous simulation without any end criterion (EndAfterIterations / '
                     'EndAfterRays / EndAfterHits are all inf and no endIf callback)')
  sources = _bake.lightSources(doc)
  if not sources:
    raise ValueError('document has no light source')
  rpi = float(settings._props.get('RaysPerIteration', 100)) if settings is not None else 100.0
  enabled = enabledHitMetadata(settings)
  own = tracer is None
  tr = tracer or Tracer(device)
  if own and compileScene in ('auto', 'structure') and (continuous or compileScene == 'structure'):
    tr.compileScene(compileScene)
  keep_rows = None
  if keepOnDevice and hasattr(tr, 'archiveHits') and ranks.world == 1 and len(sources) == 1:
    results_store.releaseDeviceRuns()                    # (one run at a time keeps its rows on the device)
    # the archive lives on a context of its own: launches' fetch threads append to it while the main thread launches on
    # the tracing contexts (a context is not to be used from two threads at once)
    budget_gb = _keep_budget_gb(tr, keepOnDevice)
    if budget_gb > 0:
      arch = Tracer(tr.device, referenceStrict=getattr(tr, 'referenceStrict', None))
      keep_rows = dict(tracer=arch, budget=int(budget_gb * 1e9 // 64), rows=0, complete=True, lock=threading.Lock())
  master = ranks.rank == 0
  if master:
    store.setStatus('simulation-is-done', False)
    store.setStatus('simulation-is-canceled', False)
    store.setStatus('simulation-is-running', True)
  failed = True
  try:
    if master:
      store.dumpGlobalInfo(_bake.collectGlobalInfo(doc))
    baked = []
    for src in sources:
      scene = _bake.bakeScene(doc, src)
      baked.append((src, scene, bakeLightSource(doc, src, seed), _bake.bakeLimits(doc, src, **traceKwargs)))
    first = {src.Name: 0 for src in sources}
    ended = False
    uploaded, hits_per_ray = {}, {}
    if (overlapFetch and own and continuous and not pseudo and len(baked) == 1 and ranks.world == 1 and endIf is None
        and not np.isfinite(store.endAfterHits) and hasattr(tr, 'hitColumns')
        and not isinstance(baked[0][2], replay_source.BakedReplay) and not baked[0][0]._props.get('RecordRays', False)):
      _run_overlapped(store, tr, baked[0], rpi, raysPerLaunch, seed, enabled, compileScene, keep_rows)
      baked = []                        # (the loop below has nothing left to do)
    while baked:
      for src, scene, bsrc, lim in baked:
        per_iter = max(1, int(round(rpi * bsrc.rays_per_iteration_scale)))
        base = first[src.Name]
        # RecordRays (generic_source.py:26, 78-118): every segment of every ray is kept, so
        # launches of such a source stay small
        record_rays = bool(src._props.get('RecordRays', False))
        rpl = min(raysPerLaunch, _RECORD_RAYS_PER_LAUNCH) if record_rays else raysPerLaunch
        # -- what this launch traces: device-generated rays or explicit initial conditions ----
        explicit = None          # (origins, directions, powers or None, wavelengths or None)
        if isinstance(bsrc, replay_source.BakedReplay):
          # replay_source.py:116-166: no fans; true and pseudo modes alike take the next
          # RaysPerIteration rays of the stock; an exhausted stock ends the simulation
          if action in ('fans', 'singlefans'):
            continue
          iters = 1 if not continuous else _iterations_for_launch(store, per_iter, rpl)
          o, d, wl, pw = bsrc.take(iters * per_iter)
          n, iters = len(o), max(1, -(-len(o) // per_iter))
          order = np.argsort(wl, kind='stable')          # one launch per wavelength (gratings)
          explicit = (o[order], d[order], pw[order], wl[order])
          per_ray = dict(initPoint=explicit[0], initDirection=explicit[1], initPower=explicit[2],
                         initWavelength=explicit[3])
          if bsrc.remaining == 0:
            import warnings
            warnings.warn(f'replay light source {src.Name} ran out of rays, canceling simulation...')
            ended = True
          if not n:
            continue
        elif action in ('fans', 'singlefans'):
          surface = isinstance(bsrc, surface_source.BakedSurfaceSource)
          # surface sources: normal rays on a grid of roughly equidistant points of every face
          # (surface_source.py:467-519); point sources: fans through the optical axis
          rays = surface_fans.generateFanRays(doc, src) if surface else point_source.generateFanRays(src, bsrc)
          n, iters, base = len(rays), 1, 0
          if not n:
            continue
          explicit = (np.array([r[0] for r in rays]), np.array([r[1] for r in rays]), None, None)
          per_ray = dict(initPoint=explicit[0], initDirection=explicit[1], initPower=np.ones(n),
                         initWavelength=np.full(n, bsrc.wavelength))
          keys = ('initPhi', 'initTheta') if surface else ('fanIndex', 'rayIndex', 'totalFanCount', 'totalRaysInFan',
                                                            'initPhi', 'initTheta')
          for key in keys:
            per_ray[key] = np.array([r[2][key] for r in rays])
        elif pseudo and isinstance(bsrc, point_source.BakedSource):
          # (surface sources treat 'pseudo' like 'true', surface_source.py:521)
          iters = 1 if not continuous else min(pseudoIterationsPerLaunch,
                                               _iterations_for_launch(store, per_iter, rpl))
          vrv = point_source.getVrv(src)
          ang = np.concatenate([vrv.drawPseudo(N=per_iter) for _ in range(iters)], axis=-1)
          rays = [point_source.makeRay(bsrc, t, p) for t, p in ang.T]
          n = len(rays)
          explicit = (np.array([r[0] for r in rays]), np.array([r[1] for r in rays]), None, None)
          per_ray = dict(initPoint=explicit[0], initDirection=explicit[1], initPower=np.ones(n),
                         initWavelength=np.full(n, bsrc.wavelength), initPhi=ang[1],
                         initTheta=ang[0] if np.isfinite(bsrc.focal_length) else np.full(n, np.nan))
        else:
          iters = 1 if not continuous else _iterations_for_launch(store, per_iter, rpl)
          n = iters * per_iter
          per_ray = None
        # -- this rank's share of the launch ---------------------------------------------------
        lo, m = ranks.shard(0, n)
        # tables travel to the device only when they change (a single-source run uploads once:
        # an upload synchronises the stream and rebuilds the BVH of big scenes)
        if uploaded.get('scene') is not scene:
          tr.setScene(scene)
          uploaded['scene'] = scene
        if uploaded.get('limits') is not lim:
          tr.setLimits(lim)
          uploaded['limits'] = lim
        tr.setDetector(None)
        # Hit-list room: every intersection of every ray could be recorded (explicit launches are
        # small: reserve exactly that); for bulk launches start from what the previous launch of
        # this source needed (first launch: 4 rows per ray) and, if rows were dropped, re-trace the
        # same index range with room for what the counters say was wanted -- the trace is
        # deterministic, so the retry yields the same rays
        worst = max(16, m * (lim.max_intersections + 1))
        if explicit is not None:
          capacity = worst
        else:
          per_ray_hits = hits_per_ray.get(src.Name, 4.0)
          capacity = min(worst, int(m * per_ray_hits * 1.25) + 1024)
        if record_rays:
          tr.reserveSegments(max(16, min(m * lim.max_intersections, (1 << 31) - 1)))
        while True:
          tr.reserveHits(capacity)
          tr.reset()
          if explicit is None:
            if uploaded.get('source') is not bsrc:
              tr.setSource(bsrc)
              uploaded['source'] = bsrc
            if m:
              tr.trace(base + lo, m, seed, record_segments=record_rays)
            per_ray, index_base = _DeviceInitialConditions(tr, bsrc, base + lo, m, seed), base + lo
          else:
            o, d, pw, wl = (a[lo:lo + m] if a is not None else None for a in explicit)
            tr.setSurfaceSeed(seed)
            uploaded.pop('source', None)           # setWavelength below overrides the source's value
            index_base = base
            if wl is None:
              # every Ray carries its source's wavelength (point_source.py:459, surface_source.py:110)
              tr.setWavelength(bsrc.wavelength)
              if m:
                tr.traceRays(o, d, pw, first=base + lo, record_segments=record_rays)
            else:
              for w in np.unique(wl):
                sel = np.nonzero(wl == w)[0]                 # contiguous: the launch is ordered by wavelength
                tr.setWavelength(w)
                tr.traceRays(o[sel], d[sel], pw[sel], first=base + lo + int(sel[0]), record_segments=record_rays)
          tr.sync()
          cnt = tr.counters()
          Tracer.raiseForRayErrors(cnt)
          if not cnt['hits_dropped']:
            break
          if capacity >= worst:
            raise RuntimeError(f'{cnt["hits_dropped"]} hit rows did not fit the device buffer of {capacity} rows')
          capacity = min(worst, max(2 * capacity, int(cnt['recorded_hits'] * 1.05) + 1024))
        if m:
          hits_per_ray[src.Name] = max(cnt['recorded_hits'] / m, 0.25)
        first[src.Name] = base + n
        before = store.totalRecordedHits
        if hasattr(tr, 'hitColumns'):
          _store_hit_columns(store, tr, scene, src, per_ray, index_base, enabled)
          _keep_rows(keep_rows, tr, cnt['recorded_hits'])
        else:                                   # (test doubles without the columnar fetch)
          _store_hits(store, tr.hits(), scene, src, per_ray, index_base, enabled)
        mine = store.totalRecordedHits - before
        if record_rays:
          _, dropped = tr.segmentCount()
          if dropped:
            raise RuntimeError(f'{dropped} ray segments did not fit the device buffer')
          store.addRays(src.Name, src._props.get('Label', src.Name), segmentsToRays(tr.segments(), scene))
        (everyone,) = ranks.sum([mine])
        store.totalRecordedHits += everyone - mine         # every rank sees the job's totals
        store.incrementRayCount(n)
        store.incrementIterationCount(iters)
      store.flush(wait=False)                 # the writer thread pickles while the next launch runs
      if master:
        store.dumpProgress()
      stop = ended or not continuous or store.reachedEnd()
      if not stop and endIf is not None:
        store.drain()                           # the callback reads the run folder
        (votes,) = ranks.sum([1 if endIf(store) else 0])
        stop = votes > 0
      if stop:
        break
    store.drain(stop=True)
    ranks.barrier()
    failed = False
  finally:
    if failed:
      try:
        store.drain(stop=True)
      except Exception:
        pass
    # any exception cancels the run (simulation_loop.py:715-723)
    if master or failed:
      store.setStatus('simulation-is-canceled', failed)
      store.setStatus('simulation-is-done', not failed)
      store.setStatus('simulation-is-running', False)
    if keep_rows is not None:
      if not failed and keep_rows['complete'] and keep_rows['rows']:
        keep_rows['tracer'].scene = getattr(tr, 'scene', None)       # (group names of deviceHits('name'))
        results_store.registerDeviceRun(store, keep_rows['tracer'], True)     # (lives on with the rows: the registry closes it)
      else:
        keep_rows['tracer'].close()
    if own:
      tr.close()
  return store


def _keep_budget_gb(tr, keepOnDevice):
  """GB of hit rows a run may keep in HBM: True / 'auto' = up to 64 GB, and never more than a third of what is free now
  (other contexts, other ranks of a rehearsal on the same GPU); a number = that many GB"""
  if keepOnDevice is True or keepOnDevice == 'auto':
    want = 64.0
  else:
    return max(0.0, float(keepOnDevice))
  try:
    free, _ = tr.memInfo()
    return min(want, free / 3e9)
  except Exception:
    return want


def _keep_rows(keep_rows, t, recorded):
  """runSimulation(keepOnDevice=...): the launch's rows join the run's archive in HBM (device to device; on the archive's
  own context, one append at a time)"""
  if keep_rows is None or not keep_rows['complete'] or not recorded:
    return
  with keep_rows['lock']:
    if not keep_rows['complete']:
      return
    if keep_rows['rows'] + recorded > keep_rows['budget']:
      keep_rows['complete'] = False                       # (over the budget: the run folder is the only copy)
      keep_rows['tracer'].archiveReset()
      return
    keep_rows['rows'] = keep_rows['tracer'].archiveHits(source=t)


def _run_overlapped(store, tr, baked, rpi, raysPerLaunch, seed, enabled, compileScene, keep_rows=None):
  """the continuous loop for one device-generated source with the fetch of launches overlapping the trace of the next
  ones (runSimulation: overlapFetch).  Launch k runs on context k % 3; one of TWO fetch threads waits for it, reads its
  counters, has its rows selected and split into columns on the device, copies them into page-locked arrays and hands
  them to the store's writer threads -- while the selection and the column kernels of one launch run, the copy engine
  moves the columns of the one before --; the main thread meanwhile launches on the next context.  The store is touched
  by one thread at a time (a lock around adding a launch's rows and flushing them).  End criteria on rays and iterations
  are known at launch time, so the run traces exactly what the one-launch-at-a-time loop traces."""
  from concurrent.futures import ThreadPoolExecutor
  src, scene, bsrc, lim = baked
  n_lanes = max(2, int(os.environ.get('ODW_RUN_LANES', '3')))
  lanes = [tr] + [Tracer(tr.device, referenceStrict=tr.referenceStrict) for _ in range(n_lanes - 1)]
  store_lock = threading.Lock()
  try:
    for t in lanes[1:]:
      if compileScene in ('auto', 'structure'):
        t.compileScene(compileScene)
    for t in lanes:
      t.setScene(scene)
      t.setLimits(lim)
      t.setSource(bsrc)
      t.setDetector(None)
    per_iter = max(1, int(round(rpi * bsrc.rays_per_iteration_scale)))
    worst_per_ray = lim.max_intersections + 1
    state = dict(hits_per_ray=4.0)

    clock = dict(wait=0.0, columns=0.0, keep=0.0, launches=0) if os.environ.get('ODW_RUN_TIMING') else None

    def fetch(t, base, n, iters, capacity):
      t0 = time.perf_counter()
      while True:
        t.sync()
        cnt = t.counters()
        Tracer.raiseForRayErrors(cnt)
        if not cnt['hits_dropped']:
          break
        worst = max(16, n * worst_per_ray)
        if capacity >= worst:
          raise RuntimeError(f'{cnt["hits_dropped"]} hit rows did not fit the device buffer of {capacity} rows')
        capacity = min(worst, max(2 * capacity, int(cnt['recorded_hits'] * 1.05) + 1024))
        t.reserveHits(capacity)                        # (deterministic: the same rays again, with room)
        t.reset()
        t.trace(base, n, seed)
      state['hits_per_ray'] = max(cnt['recorded_hits'] / n, 0.25)
      per_ray = _DeviceInitialConditions(t, bsrc, base, n, seed)
      t1 = time.perf_counter()
      _store_hit_columns(store, t, scene, src, per_ray, base, enabled, lock=store_lock, flush=True)
      t2 = time.perf_counter()
      _keep_rows(keep_rows, t, cnt['recorded_hits'])
      if clock is not None:
        t3 = time.perf_counter()
        clock['wait'] += t1 - t0; clock['columns'] += t2 - t1; clock['keep'] += t3 - t2; clock['launches'] += 1

    pending = [None] * n_lanes
    base, k = 0, 0
    with ThreadPoolExecutor(max_workers=min(2, n_lanes - 1), thread_name_prefix='odw-hit-fetch') as pool:
      try:
        while True:
          lane = k % n_lanes
          if pending[lane] is not None:
            pending[lane].result()                     # this context's rows are out (errors of its fetch surface here)
            pending[lane] = None
          t = lanes[lane]
          iters = _iterations_for_launch(store, per_iter, raysPerLaunch)
          n = iters * per_iter
          capacity = min(max(16, n * worst_per_ray), int(n * state['hits_per_ray'] * 1.25) + 1024)
          t.reserveHits(capacity)
          t.reset()
          t.trace(base, n, seed)
          pending[lane] = pool.submit(fetch, t, base, n, iters, capacity)
          base += n
          k += 1
          with store_lock:
            store.incrementRayCount(n)
            store.incrementIterationCount(iters)
            store.dumpProgress()
            if store.reachedEnd():
              break
      finally:
        errors = []
        for f in pending:
          if f is not None:
            try:
              f.result()
            except BaseException as e:              # (the first one is raised below, after all contexts are idle)
              errors.append(e)
        if errors:
          raise errors[0]
  finally:
    if clock is not None and clock['launches']:
      import sys
      print('[odw run timing] ms per launch in the fetch threads: ' + ', '.join(f'{k} {1e3 * v / clock["launches"]:.2f}' for k, v in clock.items()
                                                                              if k != 'launches') +
            ' | of columns: ' + ', '.join(f'{k} {1e3 * v / clock["launches"]:.2f}' for k, v in (_STORE_CLOCK or {}).items()), file=sys.stderr, flush=True)
      for k in (_STORE_CLOCK or {}):
        _STORE_CLOCK[k] = 0.0
    for t in lanes[1:]:
      t.close()


def bakeLightSource(doc, src, seed=0):
  """device-side description of a light source by proxy class"""
  cls = src.ProxyClass
  if cls == 'ReplaySourceProxy':
    return replay_source.bakeReplaySource(doc, src, seed)
  if cls == 'SurfaceSourceProxy':
    return surface_source.bakeSurfaceSource(doc, src)
  return point_source.bakeSource(doc, src)


def _iterations_for_launch(store, per_iter, raysPerLaunch):
  """iterations to fuse into the next launch: up to the nearest end criterion
  (+1 iteration, the criteria are strict '>'), at most raysPerLaunch rays"""
  cap = max(1, int(raysPerLaunch // per_iter))
  need = cap
  if np.isfinite(store.endAfterRays):
    need = min(need, int((store.endAfterRays - store.totalTracedRays) // per_iter) + 1)
  if np.isfinite(store.endAfterIterations):
    need = min(need, int(store.endAfterIterations - store.totalIterations) + 1)
  return max(1, need)


_METADATA_KEYS = ('initPoint', 'initDirection', 'initPower', 'initWavelength', 'initPhi', 'initTheta', 'rayIndex',
                  'fanIndex', 'totalFanCount', 'totalRaysInFan')


def enabledHitMetadata(settings):
  """metadata keys stored with every hit: the `StoreHit<Key>` switches of the
  active simulation settings (ray.py:55-65, simulation_settings.py:55-76);
  without a settings object nothing is stored (ray.py:74-75)"""
  if settings is None:
    return ()
  on = {k[8:].lower() for k, v in settings._props.items() if k.startswith('StoreHit') and v}
  return tuple(k for k in _METADATA_KEYS if k.lower() in on)


class _DeviceInitialConditions:
  """initial conditions of device-generated rays, recomputed on demand from the
  counter-based stream (the reference carries them along in Ray.metadata)"""

  def __init__(self, tracer, bsrc, first, n, seed):
    self._tr, self._src, self._first, self._n, self._seed = tracer, bsrc, first, n, seed
    self._cache = {}

  def __contains__(self, key):
    if key in ('initPoint', 'initDirection', 'initPower', 'initWavelength'):
      return True
    return key in ('initPhi', 'initTheta') and isinstance(self._src, point_source.BakedSource)

  def __getitem__(self, key):
    if key in ('initPoint', 'initDirection'):
      if 'rays' not in self._cache:
        self._cache['rays'] = self._tr.generateRays(self._first, self._n, self._seed)
      return self._cache['rays'][0 if key == 'initPoint' else 1]
    if key in ('initPhi', 'initTheta'):
      if 'angles' not in self._cache:
        self._cache['angles'] = self._tr.sample(self._first, self._n, self._seed)
      t, phi = self._cache['angles']
      if key == 'initPhi':
        return phi
      return t if np.isfinite(self._src.focal_length) else np.full(self._n, np.nan)
    if key == 'initPower':
      return np.full(self._n, self._src.power)
    if key == 'initWavelength':
      return np.full(self._n, self._src.wavelength)
    raise KeyError(key)


_STORE_CLOCK = {'lock': 0.0, 'add+flush': 0.0} if os.environ.get('ODW_RUN_TIMING') else None


def _store_hit_columns(store, tr, scene, src, per_ray, base, enabled, lock=None, flush=False):
  """the launch's rows into the store, one recording group at a time, as the device hands them over: already
  split into the arrays of the reference's hit dictionary (Tracer.hitColumns) -- the host copies nothing.
  lock: held while the store is touched (the overlapped loop's fetch threads); flush: hand the rows to the writer
  threads at once, under the same lock"""
  keys = [k for k in enabled if k in per_ray]
  columns = {k: per_ray[k] for k in keys}
  # Page-locked destination arrays (the copy engine writes them directly: 45+ GB/s instead of 17) when the rows
  # only pass through the host on their way into the run folder: the writer thread drops them after pickling and
  # their memory is used again a few launches later.  Rows that are KEPT in memory would pin a new slab per launch
  # (page-locking costs more than the staged copy saves: measured 4.0e7 against 1.1e8 rays/s): plain arrays.
  pinned = store.basePath is not None and not getattr(store, 'keepInMemory', False)
  got = []
  for g in np.nonzero(np.asarray(scene.group_record))[0]:
    cols = tr.hitColumns(int(g), pinned=pinned, rayIndex=bool(columns))       # (the ray's number: for per-ray metadata only)
    if cols is None:
      continue
    extra = {}
    if columns:
      ray = cols['rayIndex'] - int(base)
      extra = {k: np.asarray(v)[ray] for k, v in columns.items()}
    got.append((g, cols, extra))
  import contextlib
  t_a = time.perf_counter()
  with (lock if lock is not None else contextlib.nullcontext()):
    t_b = time.perf_counter()
    for g, cols, extra in got:
      store.addRayHits(src.Name, src._props.get('Label', src.Name), scene.group_names[g], scene.group_labels[g],
                       cols['points'], cols['directions'], cols['powers'], cols['isEntering'], **extra)
    if flush:
      store.flush(wait=False)
  if _STORE_CLOCK is not None:
    _STORE_CLOCK['lock'] += t_b - t_a
    _STORE_CLOCK['add+flush'] += time.perf_counter() - t_b


def _store_hits(store, rows, scene, src, per_ray, base, enabled):
  tags = rows['tag']
  grp = ((tags >> np.uint64(48)) & np.uint64(0x7FFF)).astype(np.int64)
  ray = (tags & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64) - int(base)
  keys = [k for k in enabled if k in per_ray]
  columns = {k: per_ray[k] for k in keys}
  for g in np.unique(grp):
    sel = grp == g
    extra = {k: np.asarray(v)[ray[sel]] for k, v in columns.items()}
    store.addRayHits(src.Name, src._props.get('Label', src.Name), scene.group_names[g],
                     scene.group_labels[g], rows['point'][sel], rows['direction'][sel],
                     rows['power'][sel], (tags[sel] >> np.uint64(63)).astype(np.int64), **extra)
