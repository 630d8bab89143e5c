#!/usr/bin/env python3
"""Randomised check of batch launches (ABI v9): random scenes (tests/random_scenes.py) in K variants that differ in
their numbers only -- every length and position scaled by 1 - k * 1e-3, the same kinds, rotations, optical types --
under one random point source; ONE launch for the K scenes against K launches of their own: every row, bit for bit,
and the counters.  Device against device (the single launches are held to the oracle by the other campaigns).
  python tests/fuzz_batch.py [scenes] [rays] [seed] [rich: 0 plain / 1 gratings, absorbing media, sequential mode /
                                                     4 paraboloids]          (ODW_COMPILE=structure: compiled kernels)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

from freecad.optics_design_workbench_amd import _native
from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import Document
from freecad.optics_design_workbench_amd.scene.placement import Placement
from freecad.optics_design_workbench_amd.simulation.simulation_loop import bakeLightSource
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
import random_scenes


class Shrunk(np.random.RandomState):
  """the stream of RandomState(seed) with every continuous draw scaled by 1 - eps: the decisions a scene is built
  from (kinds, booleans, optical types: rand, choice, randint) stay, its numbers move"""

  def __init__(self, seed, eps):
    super().__init__(seed)
    self.eps = float(eps)

  def uniform(self, low=0.0, high=1.0, size=None):
    return super().uniform(low, high, size) * (1.0 - self.eps)

  def normal(self, loc=0.0, scale=1.0, size=None):
    return super().normal(loc, scale, size) * (1.0 - self.eps)


def variants(seed, k, rich=False, paraboloids=False):
  """k scenes of one structure (None if the generator declines the seed), their limits, the first one's targets"""
  out = []
  for j in range(k):
    sc, lim, targets = random_scenes.scene(Shrunk(seed, 1e-3 * j), rich, False, paraboloids)
    out.append((sc, lim, targets))
  return [o[0] for o in out], out[0][1], out[0][2]


def random_source(rs, targets):
  doc = Document()
  make.makeSimulationSettings(doc)
  target = targets[rs.randint(len(targets))]
  pos = rs.normal(0, 1, 3)
  pos = pos / np.linalg.norm(pos) * 55.0
  z = (target - pos) / np.linalg.norm(target - pos)
  x = np.cross(z, [0.3, 0.5, 0.8]); x /= np.linalg.norm(x)
  m = np.eye(4); m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, np.cross(z, x), z, pos
  src = make.makePointSource(doc, placement=Placement(matrix=m), PowerDensity='exp(-theta^2/0.05)',
                             ThetaDomain=f'0, {rs.uniform(0.2, 0.6):.3f}')
  return bakeLightSource(doc, src, 0)


def one_trial(tr, seed, k, n, rich=False, paraboloids=False):
  """-> None (skipped: outside the batch's domain) or dict(rows, differing)"""
  rs = np.random.RandomState(seed + 1)
  try:
    scenes, lim, targets = variants(seed, k, rich, paraboloids)
  except Exception:
    return None
  if any(getattr(sc, 'surface_samplers', None) for sc in scenes) or any((np.asarray(sc.prim_type) == 5).any() for sc in scenes):
    return None                                 # stochastic surfaces, facets: traced one by one by design
  bs = random_source(rs, targets)
  cap = n * (lim.max_intersections + 1)
  tr.setLimits(lim)
  tr.setSource(bs)
  tr.setDetector(None)
  try:
    tr.setSceneBatch(scenes)
  except _native.NativeError as e:
    if 'unsupported' in str(e):
      return dict(skipped=str(e)[:120])
    raise
  tr.reset()
  tr.traceBatch(3, n, seed, cap)
  tr.sync()
  cnt_batch = tr.counters()
  assert cnt_batch['hits_dropped'] == 0
  rows, _ = tr.batchRows()
  got = []
  for j in range(k):
    tr.batchSelect(j)
    got.append(tr.hits())
    assert len(got[-1]) == rows[j]
  tr.batchSelect(None)
  total = {key: 0 for key in cnt_batch}
  differing = 0
  for j, sc in enumerate(scenes):
    tr.setScene(sc)
    tr.reserveHits(cap)
    tr.reset()
    tr.trace(3, n, seed)
    tr.sync()
    c = tr.counters()
    for key in total:
      total[key] += c[key]
    want = tr.hits()
    same = len(want) == len(got[j]) and all(np.array_equal(want[f], got[j][f]) for f in want.dtype.names)
    differing += not same
  if total != cnt_batch:
    differing += 1
  distinct = len({g['point'].tobytes() for g in got})
  return dict(rows=int(sum(len(g) for g in got)), differing=int(differing), distinct_segments=distinct)


if __name__ == '__main__':
  n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
  seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
  mode = sys.argv[4] if len(sys.argv) > 4 else '0'
  done = skipped = bad = rows = 0
  with Tracer(0) as tr:
    for s in range(n_scenes):
      r = one_trial(tr, seed0 * 100003 + s, 2 + s % 5, n, rich=mode == '1', paraboloids=mode == '4')
      if r is None or 'skipped' in r:
        skipped += 1
        if r is not None and os.environ.get('ODW_FUZZ_VERBOSE'):
          print(json.dumps(dict(scene=s, **r)), flush=True)
        continue
      done += 1
      rows += r['rows']
      if r['differing']:
        bad += 1
        print(json.dumps(dict(scene=s, **r)), flush=True)
      if done % 10 == 0:
        print(json.dumps(dict(progress=done)), flush=True)
  print(json.dumps(dict(scenes=done, skipped=skipped, rays_each=n, rows=rows, differing=bad)))
  sys.exit(1 if bad else 0)
