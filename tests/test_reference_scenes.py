"""The reference's own test scenes that exercise N3 features (stochastic
surfaces, gratings, negative and infinite focal lengths), loaded from the
reduced fixtures and run through the oracle on CPU and -- marked gpu -- through
the device, with the reference tests' own acceptance criteria where it has any
(test/50-old-tests/run-simulations.py:104-115: playground records > 99 hits)."""
import numpy as np
import pytest

from conftest import project

CASES = {
  # name: (focal length, samplers [(group, kind)], min fraction of rays recorded)
  'playground': (-14.1472, [], 0.5),            # lens with the identity modification DiracDelta(theta)
  'mirror-diffuse': (np.inf, [(0, 0)], 0.15),   # parallel beam onto a cos^2 diffuse mirror
  'grating': (-100.0, [], 0.99),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_scene_bakes_and_traces(oracle, name):
  f, samplers, frac = CASES[name]
  pr = project(name)
  assert pr.source.focal_length == f
  assert [(s.group, s.kind) for s in pr.scene.surface_samplers] == samplers
  n = 1100                                       # 100 rays x 10 iterations + overshoot
  res = oracle.trace(pr.scene, pr.source, pr.limits, 0, n, 1)
  c = res['counters']
  assert c['traced_rays'] == n and c['recorded_hits'] > max(99, frac * n)
  assert c['escaped'] + c['died'] + c['capped'] == n


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_scene_parity_on_device(native_lib, oracle, name):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  pr = project(name)
  n = 100000
  with Tracer(0) as tr:
    tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
    tr.reserveHits(2 * n)
    tr.reset()
    tr.trace(0, n, 3)
    tr.sync()
    g, gc = tr.hits(), tr.counters()
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 0, n, 3, nthreads=8)
  assert gc == ref['counters']
  assert np.array_equal(g['tag'], ref['hits']['tag'])
  assert np.abs(g['point'] - ref['hits']['point']).max() < 1e-7
  assert np.abs(g['direction'] - ref['hits']['direction']).max() < 1e-9
