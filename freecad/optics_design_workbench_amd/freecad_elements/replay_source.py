"""Replay light source: rays recorded as hits by an earlier run.

Host-side mirror of `ReplaySourceProxy` (freecad_elements/replay_source.py):
  _yieldOriginDirectionWavelengthPower  :72-113  every `*-hits.pkl` below
      ReplayFromDir, files and rows in random order, each row once per run:
      (points[i], directions[i], wavelength[i] or 1, powers[i])
  _generateRays                         :116-166 the source's placement is
      applied (p1 = gpM*origin, p2 = gpM*(origin+direction)), RaysPerIteration
      rays per iteration; fan mode places no rays; when the stock is used up
      the simulation ends (SimulationEnded)
The rays are explicit initial conditions for the device (odw_trace_rays).
"""
import os
import pickle
from dataclasses import dataclass

import numpy as np

from ..scene import bake as _bake


@dataclass
class BakedReplay:
  origins: np.ndarray       # (n,3) global
  directions: np.ndarray    # (n,3) global, as recorded (normalised by the tracer)
  wavelengths: np.ndarray   # (n,)
  powers: np.ndarray        # (n,)
  name: str = ''
  label: str = ''
  rays_per_iteration_scale: float = 1.0
  consumed: int = 0

  @property
  def remaining(self):
    return len(self.origins) - self.consumed

  def take(self, n):
    """the next (at most n) rays of the shuffled stock"""
    a, b = self.consumed, min(len(self.origins), self.consumed + int(n))
    self.consumed = b
    return self.origins[a:b], self.directions[a:b], self.wavelengths[a:b], self.powers[a:b]

  def rewind(self):
    """onInitializeSimulation: the whole stock is available again"""
    self.consumed = 0


def loadReplayRows(path):
  """-> (points, directions, wavelengths, powers) of every *-hits.pkl below path"""
  if not path:
    raise RuntimeError('please set a replay directory (Data -> Optical Emission -> Replay From Dir)')
  if not os.path.exists(path):
    raise RuntimeError(f'selected replay directory does not seem to exist: {path} ')
  pts, dirs, wls, pws = [], [], [], []
  found = False
  for root, ds, fs in os.walk(path, topdown=True):
    ds.sort()
    for f in sorted(fs):
      if not f.endswith('-hits.pkl'):
        continue
      found = True
      with open(os.path.join(root, f), 'rb') as fh:
        data = pickle.load(fh)
      n = len(data['powers'])
      wl = np.ones(n)
      have = np.asarray(data.get('wavelength', []), dtype=float)
      wl[:min(n, len(have))] = have[:n]
      pts.append(np.asarray(data['points'], dtype=float).reshape(n, 3))
      dirs.append(np.asarray(data['directions'], dtype=float).reshape(n, 3))
      wls.append(wl)
      pws.append(np.asarray(data['powers'], dtype=float).reshape(n))
  if not found:
    raise RuntimeError(f'selected replay directory does not seem to contain any ray hit datafile: {path} ')
  return np.concatenate(pts), np.concatenate(dirs), np.concatenate(wls), np.concatenate(pws)


def bakeReplaySource(doc, obj, seed=0):
  """stock of `obj` in global coordinates, shuffled with `seed` (the
  reference shuffles with Python's global `random`)"""
  try:
    p, d, wl, pw = loadReplayRows(obj._props.get('ReplayFromDir', ''))
  except RuntimeError as e:
    raise RuntimeError(f'light source {obj.Name}: {e}') from None
  gp = _bake.globalPlacements(doc, obj)[0]
  m = np.asarray(gp.rows12()).reshape(3, 4)
  p1 = p @ m[:, :3].T + m[:, 3]
  p2 = (p + d) @ m[:, :3].T + m[:, 3]
  order = np.random.RandomState(int(seed) % (1 << 32)).permutation(len(p))
  return BakedReplay(origins=p1[order], directions=(p2 - p1)[order], wavelengths=wl[order], powers=pw[order],
                     name=obj.Name, label=obj._props.get('Label', obj.Name),
                     rays_per_iteration_scale=float(obj._props.get('RaysPerIterationScale', 1)))
