"""The reference's fan-mode acceptance notebook restated on the oracle:
test/70-point-source-slow/source-and-absorber.OpticsDesign/notebooks/0-test-fan-mode.ipynb
cells 2-9 (finite focal length, theta domains) and 11-16 (parallel beam,
radius domains): for 5 densities x 7 domains, 3 fans each, the power density
estimated from the spacing of neighbouring fan rays (`Hits.fanEstimatedPowerDensities`)
must follow the source's density: max rms error < 0.1 (both parts), median
< 1e-2 (parallel part).  runSimulation('fans') runs through the oracle-backed
Tracer double in the `not gpu` suite and through the device Tracer in the gpu suite
(`backend` fixture, tests/conftest.py)."""
import os
import warnings

import numpy as np
import pytest
import scipy.optimize
import sympy as sy

from conftest import SCENES
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import runSimulation

THETA = (['exp(-theta**2/0.01**2)', 'exp(-theta**2/0.03**2)', '1', 'cos(30*theta)**2', '2-abs(theta)'],
         ['0, .1', '-.1, 0', '-.1, .1', '-.01, .02', '-.02, -.01', '.01, .02', '.01, .03'])
RADIUS = (['exp(-r**2/1**2)', 'exp(-r**2/3**2)', '1', 'cos(r/3)**2', '20-abs(r)'],
          ['0, 10', '-10, 0', '-10, 10', '-1, 2', '-2, -1', '1.05, 2.123', '1.01, 3.321'])


def _rms_errors(doc, tracer, dens, var):
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    hits = runSimulation(doc, 'fans', tracer=tracer).hits()
    fn = sy.lambdify(var, dens)
    errs = []
    for fan, (pos, powers) in hits.fanEstimatedPowerDensities().items():
      pos, powers = pos[1:-1], powers[1:-1]
      expect = fn(np.arctan(pos / 100)) if var == 'theta' else fn(pos)
      if not hasattr(expect, '__len__'):
        expect = np.full(len(pos), float(expect))
      cost = lambda a: np.sqrt(np.mean(sorted((expect - a * powers)**2)[1:-1]))
      errs.append(cost(scipy.optimize.minimize_scalar(cost).x))
  return errs


@pytest.mark.parametrize('part', ['theta', 'radius'])
def test_fan_mode_notebook_acceptance(backend, part):
  doc = open_fcstd(os.path.join(SCENES, 'source-and-absorber.FCStd'))
  src = doc.OpticalPointSource
  tracer = backend.tracer(nthreads=2)
  dists, domains = THETA if part == 'theta' else RADIUS
  errs = []
  for dens in dists:
    for dom in domains:
      src.PowerDensity = dens
      src.PhiDomain = '0, 2*pi'
      src.Fans = 3
      if part == 'theta':
        src.FocalLength, src.ThetaDomain, src.RaysPerFan = '0', dom, 50
      else:
        src.FocalLength, src.RadiusDomain, src.RaysPerFan = 'inf', dom, 70
      errs += _rms_errors(doc, tracer, dens, 'theta' if part == 'theta' else 'r')
  errs = np.array(errs)
  errs = errs[np.isfinite(errs)]
  assert len(errs) >= 3 * 30
  print(part, 'median', np.median(errs), 'max', errs.max())
  assert np.median(errs) > 0
  assert errs.max() < 0.1
  if part == 'radius':
    assert np.median(errs) < 1e-2


def _astigmatic_run(f, tracer):
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fans = f.runSimulation('fans', tracer=tracer).loadHits('*')
    true = f.runSimulation('true', raysPerLaunch=1 << 16, tracer=tracer).loadHits('*')
  return fans, true


def test_astigmatic_notebook_acceptance(backend):
  """2-test-astigmatic-beams.ipynb, line for line through the FreecadDocument
  facade: (a) uniform density over three quarters of the azimuth, source
  turned about its axis (`Placement.Rotation.Angle = 180+20` -- FreeCAD takes
  radians, so this is 200 rad): which corner of the detector histogram stays
  dark pins the azimuth convention of _makeRay and the axis convention of
  Hits.histogram at once (cells 3-9); (b) astigmatic Gaussian: the fan along
  the narrow axis sees a strongly varying density, the one along the wide axis
  an almost flat one, the histogram is an ellipse along the expected diagonal
  (cells 11-18)"""
  from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument
  tracer = backend.tracer(nthreads=4)
  with FreecadDocument(os.path.join(SCENES, 'source-and-absorber.FCStd'), workInTempCopy=True) as f:
    f.OpticalSimulationSettings.EndAfterRays = 'inf'
    f.OpticalSimulationSettings.EndAfterHits = '1e5'
    s = f.OpticalPointSource
    s.Placement.Rotation.Angle = 180 + 20
    s.PowerDensity = '1'
    s.FocalLength = '0'
    s.ThetaDomain = '0, .06'
    s.PhiDomain = '0, 3/2*pi'
    s.RaysPerFan = 50
    s.Fans = 12
    fans, true = _astigmatic_run(f, tracer)
    for fan, (theta, powers) in fans.fanEstimatedPowerDensities().items():
      if fan >= 7:
        assert min(theta) >= 0
    H = true.histogram(bins=30)
    assert H.hist[-10, 8] == 0
    assert H.hist[-20, 8] > 0
    assert H.hist[-10, 11] > 0
    assert H.hist[11, 20] > 0
  with FreecadDocument(os.path.join(SCENES, 'source-and-absorber.FCStd'), workInTempCopy=True) as f:
    f.OpticalSimulationSettings.EndAfterRays = 'inf'
    f.OpticalSimulationSettings.EndAfterHits = '1e5'
    s = f.OpticalPointSource
    s.FocalLength = '0'
    s.PowerDensity = 'exp(-2*((theta*cos(phi))^2/(0.01^2) + (theta*sin(phi))^2/(0.1^2)))'
    s.ThetaDomain = '0, .06'
    s.PhiDomain = '0, 2*pi'
    s.RaysPerFan = 50
    s.Fans = 12
    fans, true = _astigmatic_run(f, tracer)
    for fan, (theta, powers) in fans.fanEstimatedPowerDensities().items():
      if fan == 6:
        assert max(powers) - min(powers) < 10
      if fan == 0:
        assert max(powers) - min(powers) > 30
    # The beam axis is the plane normal, so the automatic in-plane x axis is x or y
    # depending on which component of the fitted normal is smaller -- a 1e-3 effect of the
    # sample (hits.py:160-171).  The notebook's bins are those of the x choice: fix it.
    auto = true.histogram(bins=30)
    assert max(abs(auto._xInPlaneVec[0]), abs(auto._xInPlaneVec[1])) > 0.999
    H = true.histogram(bins=30, xInPlaneVec=np.array([1.0, 0.0, 0.0]))
    assert H.hist[15, -15] > 300
    assert H.hist[15, -25] < 200
    assert H.hist[20, -20] > 200
    assert H.hist[20, -15] < 200
    assert H.hist[10, -10] > 300
