"""Hits fan math (jupyter_utils/hits.py:227-444) against the reference's own
Hits class on synthetic fan-mode hit sets (tests/golden/fan_math.npz; inputs
are rebuilt here by tests/golden/make_golden.fan_math_inputs)."""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)


@pytest.fixture(scope='module')
def golden():
  return np.load(os.path.join(GOLDEN, 'fan_math.npz'))


@pytest.fixture(scope='module')
def inputs():
  import make_golden
  return make_golden.fan_math_inputs()


def close(a, b):
  a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
  assert a.shape == b.shape, (a.shape, b.shape)
  assert np.allclose(a, b, rtol=1e-9, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize('name', ['regular', 'caustic', 'nozero'])
def test_fan_math_matches_reference(golden, inputs, name):
  from freecad.optics_design_workbench_amd.jupyter_utils import Hits
  h = Hits({k: v.copy() for k, v in inputs[name].items()})
  assert h.supportsFanMath()
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    close(h.fanCenter(), golden[name + '_center'])
    close(h.fanCenterDists(), golden[name + '_centerDists'])
    close(h.fanNeighborDists(), golden[name + '_neighborDists'])
    close(h.fanCurvs(), golden[name + '_curvs'])
    close(h.fanMissingRays(), golden[name + '_missing'])
    close(h.fanSkippedRays(), golden[name + '_skipped'])
    assert bool(h.fanSymmetryHealthy()) == bool(golden[name + '_healthy'])
    assert h.raysPerFan() == golden[name + '_raysPerFan'] and h.fanCount() == golden[name + '_fanCount']
    dens, caus = h.fanEstimatedPowerDensities(), h.fanEstimatedCausticIntensities()
    assert sorted(int(i) for i in dens) == [0, 1]
    xs = np.linspace(-9, 9, 37)
    for i in dens:
      close(dens[i], golden[f'{name}_density_{int(i)}'])
      close(caus[i], golden[f'{name}_caustic_{int(i)}'])
      close(h.fanEstimatedPowerDensityFuncs()[i](xs), golden[f'{name}_densityfunc_{int(i)}'])
      fn = h.fanEstimatedCausticIntensityFuncs()[i]
      close([fn(-9.0, 9.0), fn(0.0, 1.0), fn(3.9, 4.1)], golden[f'{name}_causticfunc_{int(i)}'])
  if name == 'caustic':
    assert sum(len(np.atleast_2d(v).T) for v in caus.values() if np.size(v)) > 0     # the fold is seen
  assert list(h.allRayIndices(fanI=1))[:3] == [-10, -9, -8]


def test_fan_math_needs_metadata():
  from freecad.optics_design_workbench_amd.jupyter_utils import Hits
  h = Hits(dict(points=np.zeros((3, 3)), directions=np.tile([0, 0, 1.0], (3, 1)), isEntering=np.ones(3)))
  assert not h.supportsFanMath()
  with pytest.raises(ValueError):
    h.fanCount()
  with pytest.raises(ValueError):
    Hits({}).raysPerFan()
