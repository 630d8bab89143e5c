"""Committed regression vectors of the tracer (tests/golden/trace_*.npz, made
by tests/golden/make_trace_golden.py from the oracle): the oracle must keep
reproducing them bit for bit (CPU), the device within the parity tolerance
(gpu).  hugeArray is chaotic: only its first intersections are compared on
the device (see tests/test_gpu_parity.py)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, project

SCENES_ = sorted(os.path.basename(p)[len('trace_'):-4] for p in glob.glob(os.path.join(GOLDEN, 'trace_*.npz')))


def _load(scene):
  g = np.load(os.path.join(GOLDEN, f'trace_{scene}.npz'))
  return g, int(g['first']), int(g['n']), int(g['seed'])


@pytest.mark.parametrize('scene', SCENES_)
def test_oracle_reproduces_committed_vectors(oracle, scene):
  g, first, n, seed = _load(scene)
  pr = project(scene)
  fn = oracle.trace_surface if hasattr(pr.source, 'face_prim') else oracle.trace
  r = fn(pr.scene, pr.source, pr.limits, first, n, seed)
  # (the vectors hold the eight counters of ABI 4; 'grating_in_medium', added later, stays zero)
  assert [r['counters'][k] for k in oracle.CNT_NAMES[:8]] == list(g['counters']) and r['counters']['grating_in_medium'] == 0
  h = r['hits']
  assert np.array_equal(h['tag'], g['tag'])
  # same source, same compiler flags: bit for bit; another libm may differ in the last digits
  for key in ('point', 'direction', 'power'):
    assert np.allclose(h[key], g[key], rtol=0, atol=1e-11), key


@pytest.mark.gpu
@pytest.mark.parametrize('scene', SCENES_)
def test_device_reproduces_committed_vectors(native_lib, scene):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  g, first, n, seed = _load(scene)
  pr = project(scene)
  with Tracer(0) as tr:
    tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
    tr.reserveHits(n * (pr.limits.max_intersections + 1))
    tr.reset()
    tr.trace(first, n, seed)
    tr.sync()
    h, c = tr.hits(), tr.counters()
  names = ['traced_rays', 'recorded_hits', 'segments', 'escaped', 'died', 'capped', 'hist_overflow', 'hits_dropped']
  if scene == 'hugeArray':
    assert c['traced_rays'] == n and abs(c['recorded_hits'] - int(g['counters'][1])) <= 3
    return
  assert [c[k] for k in names] == list(g['counters'])
  assert np.array_equal(h['tag'], g['tag'])
  assert np.abs(h['point'] - g['point']).max() < 1e-7 and np.abs(h['direction'] - g['direction']).max() < 1e-9
  assert np.abs(h['power'] - g['power']).max() < 1e-12
