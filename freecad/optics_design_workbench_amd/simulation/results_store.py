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


class SimulationResults:

  def __init__(self, simulationType, resultsPath=None, simulationRunFolder=None,
               endAfterIterations=np.inf, endAfterRays=np.inf, endAfterHits=np.inf):
    self.simulationType = simulationType
    self.basePath = resultsPath
    self.simulationRunFolder = simulationRunFolder
    if resultsPath is not None:
      if simulationRunFolder is None:
        self.simulationRunFolder = f'raw/simulation-run-{latestRunIndex(resultsPath) + 1:06d}'
      path = self.runFolderPath()
      os.makedirs(path, exist_ok=True)
      if not any(f.startswith('uid-') for f in os.listdir(path)):
        open(os.path.join(path, f'uid-{uuid.uuid4()}'), 'w').close()
      os.makedirs(os.path.join(resultsPath, 'notebooks'), exist_ok=True)
    self.endAfterIterations = endAfterIterations
    self.endAfterRays = endAfterRays
    self.endAfterHits = endAfterHits
    self.totalIterations = 0
    self.totalTracedRays = 0
    self.totalRecordedHits = 0
    self.t0 = time.time()
    self._hits = {}          # (sourceName, sourceLabel, objName, objLabel) -> list of dicts
    self._flushed = {}       # same key -> merged dict kept for in-memory access

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
  def flush(self):
    """write buffered hits as `*-hits.pkl` (results_store.py:405-457)"""
    ms = max(int(time.time() * 1e3), getattr(self, '_lastStampMs', 0) + 1)   # one file name per flush
    self._lastStampMs = ms
    stamp = f'{ms}-pid{os.getpid()}-thread{threading.get_ident()}'
    for key, parts in self._hits.items():
      merged = {}
      for part in parts:
        for k, v in part.items():
          updateResultEntry(merged, k, v)
      keep = self._flushed.setdefault(key, {})
      for k, v in merged.items():
        updateResultEntry(keep, k, v)
      if self.basePath is not None:
        _, sourceLabel, _, objLabel = key
        folder = os.path.join(self.runFolderPath(), f'source-{sourceLabel}', f'object-{objLabel}')
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, f'{stamp}-hits.pkl'), 'wb') as f:
          pickle.dump(merged, f)
    self._hits = {}

  def hits(self, pattern='*'):
    """everything recorded so far as one `Hits` (in-memory `loadHits`)"""
    self.flush()
    result = {}
    for (src, srcLabel, obj, objLabel), d in self._flushed.items():
      rel = f'source-{srcLabel}/object-{objLabel}'
      if pattern in ('*', '**') or fnmatch.fnmatch(rel, pattern) or fnmatch.fnmatch(rel, f'*{pattern}*'):
        for k, v in d.items():
          updateResultEntry(result, k, v)
    return Hits(result)


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

  def loadHits(self, pattern='*'):
    if pattern == '*':
      pattern = '**'
    result = {}
    for p in sorted(glob.iglob(f'{self._path}/{pattern}/*-hits.pkl', recursive=True)):
      if p.startswith(f'{self._path}/progress'):
        continue
      with open(p, 'rb') as f:
        data = pickle.load(f)
      for k, v in data.items():
        updateResultEntry(result, k, v)
    return Hits(result)


def rawFolders(resultsPath):
  raw = os.path.join(resultsPath, 'raw')
  names = sorted(f for f in os.listdir(raw) if f.startswith('simulation-run-')) if os.path.isdir(raw) else []
  return [RawFolder(os.path.join(raw, n)) for n in names]


def latestRawFolder(resultsPath):
  f = rawFolders(resultsPath)
  return f[-1] if f else None
