"""Simulation results: counters, hit buffers, run-folder files.

Same on-disk contract as the reference's `SimulationResults`
(simulation/results_store.py:263-460): a run folder
`<doc>.OpticsDesign/raw/simulation-run-%06d/` with a `uid-*` marker and, per
(light source, optical object), pickled dictionaries
`source-<Label>/object-<Label>/<ms>-pid<pid>-thread<id>-hits.pkl`
{source, obj, points (M,3), directions (M,3), powers (M,), isEntering (M,)},
so the reference's own `RawFolder.loadHits` (jupyter_utils/freecad_document.py:
1485-1504) can read what this package writes and vice versa.  Hits arrive in
bulk (arrays from the device), not one Python call per hit.
"""
import fnmatch
import glob
import os
import pickle
import queue
import threading
import time
import uuid

import numpy as np

from ..jupyter_utils.hits import Hits


def resultsFolderPath(fcstdPath):
  """results_store.py:203-215 `_getFolderBase`"""
  base, fname = os.path.split(os.path.realpath(fcstdPath))
  if fname.lower().endswith('.fcstd'):
    fname = fname[:-6]
  return os.path.join(base, fname + '.OpticsDesign')


def latestRunIndex(resultsPath):
  raw = os.path.join(resultsPath, 'raw')
  best = -1
  if os.path.isdir(raw):
    for f in os.listdir(raw):
      tail = f[len('simulation-run-'):]
      if f.startswith('simulation-run-') and tail.isnumeric():
        best = max(best, int(tail))
  return best


def updateResultEntry(result, key, value):
  """merge rule of the reference (results_store.py:218-245): strings stay,
  arrays concatenate"""
  if key not in result:
    result[key] = value
  elif isinstance(value, (str, np.str_)):
    if isinstance(result[key], (str, np.str_)):
      if result[key] != value:
        result[key] = [result[key], value]
    elif value not in result[key]:
      result[key] = list(result[key]) + [value]
  elif len(result[key]) == 0:
    result[key] = value
  elif len(value):
    result[key] = np.concatenate([result[key], value], axis=0)


def _atomic_pickle(path, obj):
  tmp = f'{path}.tmp{os.getpid()}'
  with open(tmp, 'wb') as f:
    pickle.dump(obj, f)
  os.replace(tmp, path)


class SimulationResults:

  def __init__(self, simulationType, resultsPath=None, simulationRunFolder=None,
               endAfterIterations=np.inf, endAfterRays=np.inf, endAfterHits=np.inf, owner=True,
               keepInMemory=None):
    """keepInMemory: keep flushed batches in host memory (default: only when there is no results
    folder to write them to -- a continuous run with a folder holds nothing after each flush)"""
    self.simulationType = simulationType
    self.keepInMemory = (resultsPath is None) if keepInMemory is None else bool(keepInMemory)
    self.basePath = resultsPath
    self.simulationRunFolder = simulationRunFolder
    if resultsPath is not None:
      if simulationRunFolder is None:
        self.simulationRunFolder = f'raw/simulation-run-{latestRunIndex(resultsPath) + 1:06d}'
      path = self.runFolderPath()
      os.makedirs(path, exist_ok=True)
      # (several ranks share one run folder: its uid file is written by one of them)
      if owner and not any(f.startswith('uid-') for f in os.listdir(path)):
        open(os.path.join(path, f'uid-{uuid.uuid4()}'), 'w').close()
      os.makedirs(os.path.join(resultsPath, 'notebooks'), exist_ok=True)
    self.endAfterIterations = endAfterIterations
    self.endAfterRays = endAfterRays
    self.endAfterHits = endAfterHits
    self.totalIterations = 0
    self.totalTracedRays = 0
    self.totalRecordedHits = 0
    self.totalRecordedRays = 0
    self.t0 = time.time()
    self._hits = {}          # (sourceName, sourceLabel, objName, objLabel) -> list of dicts
    self._flushed = {}       # same key -> list of flushed batches (keepInMemory only)
    self._hitFiles = {}      # same key -> `*-hits.pkl` files this process wrote
    self._rays = {}          # (sourceName, sourceLabel) -> list of ray dicts (RecordRays)
    self._flushedRays = {}
    self._rayFiles = {}

  # -- global info, progress, status flags --------------------------------------
  def dumpGlobalInfo(self, info):
    """`global-info.pkl` of the run folder, written once (results_store.py:333-336)"""
    if self.basePath is None:
      return
    path = os.path.join(self.runFolderPath(), 'global-info.pkl')
    if not os.path.exists(path):
      _atomic_pickle(path, info)

  def dumpProgress(self):
    """`progress/master-%09d` summary of the run (results_store.py:508-538):
    counters + end criteria; files older than the last ten are removed"""
    if self.basePath is None:
      return
    folder = os.path.join(self.runFolderPath(), 'progress')
    os.makedirs(folder, exist_ok=True)
    idx = getattr(self, '_masterProgressDumpIdx', 0)
    _atomic_pickle(os.path.join(folder, f'master-{idx:09d}'),
                   dict(simulationType=self.simulationType, totalIterations=self.totalIterations,
                        totalTracedRays=self.totalTracedRays, totalRecordedHits=self.totalRecordedHits,
                        totalRecordedRays=self.totalRecordedRays, endAfterIterations=self.endAfterIterations,
                        endAfterRays=self.endAfterRays, endAfterHits=self.endAfterHits))
    old = os.path.join(folder, f'master-{idx - 10:09d}')
    if os.path.exists(old):
      os.remove(old)
    self._masterProgressDumpIdx = idx + 1

  def setStatus(self, name, state):
    """flag files of the results folder: simulation-is-running / -canceled / -done
    (simulation_loop.py:174-269)"""
    if self.basePath is None:
      return
    path = os.path.join(self.basePath, name)
    if state and not os.path.exists(path):
      os.makedirs(self.basePath, exist_ok=True)
      open(path, 'w').close()
    elif not state and os.path.exists(path):
      os.remove(path)

  def runFolderPath(self):
    return None if self.basePath is None else os.path.join(self.basePath, self.simulationRunFolder)

  # -- feeding ---------------------------------------------------------------
  def addRayHits(self, source, sourceLabel, obj, objLabel, points, directions, powers, isEntering,
                 **metadata):
    """bulk form of addRayHit (results_store.py:641-648)"""
    d = dict(source=source, obj=obj, points=np.asarray(points), directions=np.asarray(directions),
             powers=np.asarray(powers), isEntering=np.asarray(isEntering, dtype=np.int64))
    for k, v in metadata.items():
      d[k] = np.asarray(v)
    self._hits.setdefault((source, sourceLabel, obj, objLabel), []).append(d)
    self.totalRecordedHits += len(d['points'])

  def addRays(self, source, sourceLabel, rays):
    """bulk form of addRay + addSegment + rayComplete (results_store.py:628-639,
    232-260): `rays` = one dict(points, powers, media) per completed ray"""
    rays = list(rays)
    self._rays.setdefault((source, sourceLabel), []).extend(rays)
    self.totalRecordedRays += len(rays)

  def incrementRayCount(self, n=1):
    self.totalTracedRays += int(n)

  def incrementIterationCount(self, n=1):
    self.totalIterations += int(n)

  def reachedEnd(self):
    """end criteria with the reference's strict '>' (results_store.py:508-510)"""
    return (self.totalIterations > self.endAfterIterations or self.totalTracedRays > self.endAfterRays
            or self.totalRecordedHits > self.endAfterHits)

  def performanceDescription(self):
    dt = max(time.time() - self.t0, 1e-9)
    return f'{self.totalTracedRays / dt:.1e} rays/s, {self.totalRecordedHits / dt:.1e} recorded hits/s'

  # -- output ----------------------------------------------------------------
  # -- writing in the background ---------------------------------------------------------------
  # Pickling 4e6 hits (290 MB) takes 30 - 85 ms, a launch that produces them 1 ms and their copy out of HBM 7: the run
  # loop hands the files of a flush to WRITER THREADS (a few of them: one thread moves 7 - 10 GB/s into the page cache,
  # the copy engine delivers 45; at most two files per thread in flight) and goes on tracing; everything that reads the
  # folder, and the end of the run, waits for them (`drain`).  Protocol 5: numpy arrays go into the file without the
  # intermediate bytes copy of the default protocol -- the file write is one system call per array, outside the
  # interpreter lock --; `pickle.load` of the reference reads them all the same.
  @staticmethod
  def _usable_cores():
    """cores this process may use: its affinity mask cut by the cgroup's CPU quota (os.cpu_count() is the HOST's count:
    256 on a box that gives a one-GPU job 16)"""
    try:
      n = len(os.sched_getaffinity(0))
    except AttributeError:
      n = os.cpu_count() or 2
    try:
      with open('/sys/fs/cgroup/cpu.max') as f:
        quota, period = f.read().split()[:2]
      if quota != 'max':
        n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
      pass
    return n

  def _writer_put(self, path, obj):
    if getattr(self, '_writers', None) is None:
      # half the usable cores write files (a thread moves ~5 GB/s into the page cache; 8 threads: 40 GB/s on the pool's
      # boxes, scripts/bench_file_write.py), at most 8; the queue holds one more launch per two writers
      n = int(os.environ.get('ODW_WRITER_THREADS', max(1, min(8, self._usable_cores() // 2))))
      self._writeQueue = queue.Queue(maxsize=max(2, n // 2))
      self._writeError = None
      try:
        # (the page-locked slabs of the launches in flight between device and files stay in the pool: tracer.py)
        from .tracer import _POOL
        _POOL.reserve(n + max(2, n // 2) + 3)
      except Exception:
        pass

      def work():
        while True:
          item = self._writeQueue.get()
          try:
            if item is None:
              return
            if self._writeError is None:
              # (written under a temporary name: a reader of the folder -- loadHits of a concurrent
              #  notebook, an endIf callback -- never meets half a pickle)
              tmp = item[0] + '.tmp'
              with open(tmp, 'wb') as f:
                pickle.dump(item[1], f, protocol=pickle.HIGHEST_PROTOCOL)
              os.replace(tmp, item[0])
          except BaseException as e:                 # raised by the next flush() / drain()
            self._writeError = e
          finally:
            self._writeQueue.task_done()
      self._writers = [threading.Thread(target=work, name=f'odw-hit-writer-{k}', daemon=True) for k in range(n)]
      for t in self._writers:
        t.start()
    self._raiseWriteError()
    self._writeQueue.put((path, obj))

  def _raiseWriteError(self):
    """a failed write (disk full, permissions) ends the run at the next flush, not at its end: the files after
    it would be skipped, and the counters would report hits that are not on disk"""
    if getattr(self, '_writeError', None) is not None:
      e, self._writeError = self._writeError, None
      raise e

  def drain(self, stop=False):
    """wait until every file handed to the writer thread is on disk; raises what the writer met.
    stop: the thread ends as well (the end of a run; a later flush starts a new one)"""
    if getattr(self, '_writers', None) is not None:
      self._writeQueue.join()
      if stop:
        for _ in self._writers:
          self._writeQueue.put(None)
        for t in self._writers:
          t.join()
        self._writers = None
      self._raiseWriteError()

  def flush(self, wait=True):
    """write buffered hits as `*-hits.pkl` and empty the buffers (results_store.py:405-457).
    With a results folder nothing stays in host memory after the write (the reference clears its
    lists, :455-457; `hits()` reads the run folder back); without one the batches are kept as a
    list of chunks and merged lazily by `hits()`.  wait=False (the run loop): the files are written by
    the writer thread while the next launch runs; `drain()` -- called by everything that reads them --
    waits for them."""
    ms = max(int(time.time() * 1e3), getattr(self, '_lastStampMs', 0) + 1)   # one file name per flush
    self._lastStampMs = ms
    stamp = f'{ms}-pid{os.getpid()}-thread{threading.get_ident()}'
    for key, parts in self._hits.items():
      merged = {}
      for part in parts:
        for k, v in part.items():
          updateResultEntry(merged, k, v)
      if self.basePath is not None:
        _, sourceLabel, _, objLabel = key
        folder = os.path.join(self.runFolderPath(), f'source-{sourceLabel}', f'object-{objLabel}')
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, f'{stamp}-hits.pkl')
        self._writer_put(path, merged)
        self._hitFiles.setdefault(key, []).append(path)
      if self.keepInMemory:
        self._flushed.setdefault(key, []).append(merged)
    self._hits = {}
    # recorded rays: one pickled list of ray dictionaries per source (results_store.py:380-403)
    for key, rays in self._rays.items():
      if self.basePath is not None and rays:
        folder = os.path.join(self.runFolderPath(), f'source-{key[1]}')
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, f'{stamp}-rays.pkl')
        self._writer_put(path, rays)
        self._rayFiles.setdefault(key, []).append(path)
      if self.keepInMemory:
        self._flushedRays.setdefault(key, []).extend(rays)
    self._rays = {}
    if wait:
      self.drain()

  @staticmethod
  def _matches(rel, pattern):
    return pattern in ('*', '**') or fnmatch.fnmatch(rel, pattern) or fnmatch.fnmatch(rel, f'*{pattern}*')

  def rays(self, pattern='*'):
    """every recorded ray of THIS process so far (in-memory `loadRays`; read back from this
    process' own `*-rays.pkl` files when the store does not keep flushed batches)"""
    self.flush()
    out = []
    for key in list(self._flushedRays) + [k for k in self._rayFiles if k not in self._flushedRays]:
      if not self._matches(f'source-{key[1]}', pattern):
        continue
      if self.keepInMemory:
        out.extend(self._flushedRays.get(key, []))
      else:
        for path in self._rayFiles.get(key, []):
          with open(path, 'rb') as f:
            out.extend(pickle.load(f))
    return out

  def deviceHits(self, group=None):
    """the run's rows where they are, in HBM (`runSimulation(keepOnDevice=True)`): a `DeviceHits`, or None if the run
    did not keep them (not asked for, over the budget, several sources or ranks)"""
    key = getattr(self, '_deviceRunKey', None)
    return None if key is None else deviceHitsOfRun(key, group)

  def hits(self, pattern='*'):
    """everything THIS process recorded so far as one `Hits` (in-memory `loadHits`)"""
    self.flush()
    result = {}
    for key in list(self._flushed) + [k for k in self._hitFiles if k not in self._flushed]:
      (src, srcLabel, obj, objLabel) = key
      if not self._matches(f'source-{srcLabel}/object-{objLabel}', pattern):
        continue
      if self.keepInMemory:
        chunks = self._flushed.get(key, [])
      else:
        chunks = []
        for path in self._hitFiles.get(key, []):
          with open(path, 'rb') as f:
            chunks.append(pickle.load(f))
      for d in chunks:
        for k, v in d.items():
          updateResultEntry(result, k, v)
    return Hits(result)


# ---- runs whose rows are still in HBM (runSimulation(keepOnDevice=True)) --------------------------------------------
_DEVICE_RUNS = {}          # real path of the run folder (or id of the store) -> (store, tracer, owned)


def registerDeviceRun(store, tracer, owned):
  key = os.path.realpath(store.runFolderPath()) if store.basePath is not None else id(store)
  _DEVICE_RUNS[key] = (store, tracer, owned)
  store._deviceRunKey = key


def releaseDeviceRuns():
  """drop the rows earlier runs kept on the device (their tracers are closed if the runs created them)"""
  for key, (store, tracer, owned) in list(_DEVICE_RUNS.items()):
    try:
      tracer.archiveReset()
      if owned:
        tracer.close()
    except Exception:
      pass
    _DEVICE_RUNS.pop(key, None)


def deviceHitsOfRun(key, group=None):
  """`DeviceHits` on the rows a run kept in HBM, or None"""
  entry = _DEVICE_RUNS.get(key)
  if entry is None:
    return None
  store, tracer, _ = entry
  store.drain()
  tracer.archiveSelect(True)
  from .device_hits import DeviceHits
  return DeviceHits(tracer, group)


class RunHits(Hits):
  """The `Hits` of a run folder.  Its arrays are read from the `*-hits.pkl` files when something first asks for them
  (`hits`, `points()`, `plot()` ...); while the rows of the run are still in HBM -- this process made the run and kept
  them (`runSimulation(keepOnDevice=...)`, the default) -- `len()`, `detectPlaneNormal()` and `histogram()` of points or
  directions work on them THERE (`DeviceHits`: plane search on the thinned sample, projection, medians and binning on the
  device, numpy's rules) without reading anything back: 0.15 s instead of 6.5 s for a 5e7-hit run."""

  def __init__(self, loader, deviceKey=None):
    self._loader, self._loaded, self._deviceKey = loader, None, deviceKey

  @property
  def hits(self):
    if self._loaded is None:
      self._loaded = self._loader()
    return self._loaded

  @hits.setter
  def hits(self, value):
    self._loaded = value

  def _device(self):
    """DeviceHits on the run's rows in HBM, or None (the run kept none, another run has taken their place, the arrays
    were read already)"""
    if self._deviceKey is None or self._loaded is not None:
      return None
    try:
      return deviceHitsOfRun(self._deviceKey)
    except Exception:
      return None

  def __len__(self):
    dev = self._device()
    return len(dev) if dev is not None else super().__len__()

  def detectPlaneNormal(self, points=None, directions=None, planeNormal=None, xInPlaneVec=None, maxPointCountConsidered=300,
                        angleTol=1e-9):
    dev = self._device() if (points is None and directions is None) else None
    if dev is not None and len(dev):
      return dev.detectPlaneNormal(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec, maxPointCountConsidered=maxPointCountConsidered,
                                   angleTol=angleTol)
    return super().detectPlaneNormal(points=points, directions=directions, planeNormal=planeNormal, xInPlaneVec=xInPlaneVec,
                                     maxPointCountConsidered=maxPointCountConsidered, angleTol=angleTol)

  def histogram(self, planeNormal=None, xInPlaneVec=None, key='points', **kwargs):
    dev = self._device() if key in ('points', 'directions') else None
    if dev is not None and len(dev):
      try:
        return dev.histogram(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec, key=key, **kwargs)
      except TypeError:                       # (an argument the device route does not take: the arrays, numpy)
        pass
    return super().histogram(planeNormal=planeNormal, xInPlaneVec=xInPlaneVec, key=key, **kwargs)

  def deviceHits(self):
    """the `DeviceHits` behind this object, or None"""
    return self._device()


class RawFolder:
  '''
  One simulation-run folder of the raw results directory
  (jupyter_utils/freecad_document.py:1393-1504).
  '''

  def __init__(self, path):
    self._path = path
    uids = [f for f in os.listdir(path) if f.startswith('uid') and os.path.isfile(os.path.join(path, f))]
    if len(uids) != 1:
      raise RuntimeError('invalid raw data folder: ' + ('uid file missing' if not uids else 'more than one uid file'))
    self._uid = uids[0][4:]

  def __repr__(self):
    return f'<RawFolder {os.path.basename(self._path)}/ UID={self._uid}>'

  def path(self):
    return os.path.relpath(self._path)

  def tree(self, _path=None):
    """nested dictionaries of the folder structure; files are summarised as
    '<n hit files>' / '<n ray files>' / '<n unknown files>' entries (freecad_document.py:1437-1463)"""
    path = _path or self._path
    result, kinds = {}, []
    for d in os.scandir(path):
      if d.is_dir():
        result[d.name] = self.tree(d.path)
      elif d.name.endswith('-hits.pkl'):
        kinds.append('hit')
      elif d.name.endswith('-rays.pkl'):
        kinds.append('ray')
      elif not (d.name.startswith('uid-') or d.name.startswith('.')):
        kinds.append('unknown')
    for k in set(kinds):
      result[f'<{kinds.count(k)} {k} files>'] = None
    return result

  def printTree(self, _node=None, _prefix='  '):
    if _node is None:
      print(f'{os.path.basename(self._path)}/')
      _node = self.tree()
    for k, v in sorted(_node.items()):
      if v is None:
        print(_prefix + k)
      else:
        print(_prefix + k + '/')
        self.printTree(v, _prefix + '  ')

  def reload(self):
    pass                       # nothing is cached here

  def loadGlobalInfo(self):
    with open(os.path.join(self._path, 'global-info.pkl'), 'rb') as f:
      return pickle.load(f)

  def loadProgress(self):
    """the latest `progress/master-*` summary (None if the run wrote none)"""
    files = sorted(glob.glob(os.path.join(self._path, 'progress', 'master-*')))
    if not files:
      return None
    with open(files[-1], 'rb') as f:
      return pickle.load(f)

  def loadHits(self, pattern='*', device=None):
    """every hit of the run as `Hits` (freecad_document.py:1485-1504).  The arrays are read from the files when they are
    first asked for; if this process made the run and its rows are still in HBM (`runSimulation(keepOnDevice=...)`, the
    default), `len()`, `detectPlaneNormal()` and `histogram()` of the returned object work on them there (`RunHits`) --
    every recording group's rows, so only for pattern '*'.  device=True: the `DeviceHits` itself if the rows are there
    (histogram(), detectPlaneNormal(), moments() ...; `toHits()` for the arrays); device=False: never look at the device."""
    everything = pattern in ('*', '**')
    key = os.path.realpath(self._path)
    if device and everything:
      hits = deviceHitsOfRun(key)
      if hits is not None:
        return hits
    if pattern == '*':
      pattern = '**'

    def load(pattern=pattern):
      result = {}
      for p in sorted(glob.iglob(f'{self._path}/{pattern}/*-hits.pkl', recursive=True)):
        if p.startswith(f'{self._path}/progress'):
          continue
        with open(p, 'rb') as f:
          data = pickle.load(f)
        for k, v in data.items():
          updateResultEntry(result, k, v)
      return result

    if device is None and everything and key in _DEVICE_RUNS:
      return RunHits(load, deviceKey=key)
    return Hits(load())

  def loadRays(self, pattern='*'):
    """the rays of sources with RecordRays: a list of dictionaries
    (points (k+1, 3), powers (k,), media [k]), one per ray, as
    SimulationResults.flush pickles them (results_store.py:380-399).  (The
    reference's own loadRays, freecad_document.py:1488-1504, merges the files
    like hit dictionaries and so rejects these lists; here they are
    concatenated.)"""
    if pattern == '*':
      pattern = '**'
    out = []
    for p in sorted(glob.iglob(f'{self._path}/{pattern}/*-rays.pkl', recursive=True)):
      with open(p, 'rb') as f:
        out.extend(pickle.load(f))
    return out


def _rawBase(basePath='.'):
  """the `raw` folder a path belongs to (freecad_document.py:1341-1360): a single
  `*.OpticsDesign` folder below basePath is entered, otherwise the parents are searched"""
  basePath = os.path.abspath(basePath)
  sims = [p for p in os.listdir(basePath) if p.endswith('.OpticsDesign') and os.path.isdir(os.path.join(basePath, p))] \
      if os.path.isdir(basePath) else []
  if len(sims) == 1 and not os.path.exists(os.path.join(basePath, 'raw')):
    basePath = os.path.join(basePath, sims[0])
  while not os.path.exists(os.path.join(basePath, 'raw')) and basePath != os.path.dirname(basePath):
    basePath = os.path.dirname(basePath)
  raw = os.path.join(basePath, 'raw')
  if not os.path.exists(raw):
    raise ValueError(f'failed to find "raw" folder in any parent directory of {basePath!r}')
  folders = sorted(d for d in os.listdir(raw) if d.startswith('simulation-run-'))
  return raw, folders, [int(d[len('simulation-run-'):]) for d in folders]


class RawFolderRange:
  """several run folders at once (freecad_document.py:1506-1539): iterable, sliceable,
  `loadHits` / `loadRays` merge the folders' results"""

  def __init__(self, paths):
    self._paths = [p._path if isinstance(p, RawFolder) else p for p in paths]

  def __iter__(self):
    return iter([RawFolder(p) for p in self._paths])

  def __len__(self):
    return len(self._paths)

  def __bool__(self):
    return bool(self._paths)

  def __getitem__(self, i):
    sel = self._paths[i]
    return RawFolder(sel) if isinstance(sel, str) else RawFolderRange(sel)

  def paths(self):
    return [os.path.relpath(p) for p in self._paths]

  def loadHits(self, pattern='*'):
    result = {}
    for r in self:
      for k, v in r.loadHits(pattern).items():
        updateResultEntry(result, k, v)
    return Hits(result)

  def loadRays(self, pattern='*'):
    return [ray for r in self for ray in r.loadRays(pattern)]


def rawFolders(basePath='.'):
  """RawFolderRange of all run folders under basePath (a results folder, a project folder with one
  `.OpticsDesign` folder, or anything below a `raw` folder)"""
  try:
    raw, folders, _ = _rawBase(basePath)
  except ValueError:
    return RawFolderRange([])
  return RawFolderRange([os.path.join(raw, f) for f in folders])


def rawFolderByIndex(index=-1, basePath='.'):
  """index >= 0: the run folder of that number; negative: counted from the latest"""
  raw, folders, indices = _rawBase(basePath)
  if index >= 0:
    if index not in indices:
      raise ValueError(f'simulation-run folder with index {index} does not exist')
    return RawFolder(os.path.join(raw, folders[indices.index(index)]))
  return RawFolder(os.path.join(raw, folders[index]))


def latestRawFolder(basePath='.'):
  f = rawFolders(basePath)
  return f[-1] if len(f) else None
