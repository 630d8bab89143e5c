import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SCENES = os.path.join(GOLDEN, 'scenes')


# the sampler's analytic attempt waits up to 2 s for sympy (like the reference's); the suites compile dozens of densities it
# never inverts -- the same outcome after 0.4 s (tests of the analytic mode itself pass their own timeouts)
os.environ.setdefault('ODW_ANALYTIC_TIMEOUT', '0.4')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def native_lib():
  """the HIP library, built in-tree if needed; never a fallback"""
  from freecad.optics_design_workbench_amd import _native
  _native.build()
  return _native.lib()


@pytest.fixture(scope='session')
def oracle():
  from oracle import capi
  capi.build()
  return capi


_PROJECTS = {}


def project(name, **kw):
  from freecad.optics_design_workbench_amd import scenes
  key = (name, tuple(sorted(kw.items())))
  if key not in _PROJECTS:
    _PROJECTS[key] = scenes.bakeProject(os.path.join(SCENES, name + '.FCStd'), **kw)
  return _PROJECTS[key]


# ---------------------------------------------------------------------------
# The reference-held acceptance criteria run twice: on the CPU oracle (`not gpu`
# suite) and on the HIP path itself (`-m gpu`), through the same test bodies.
BACKENDS = ['oracle', pytest.param('device', marks=pytest.mark.gpu)]


class _Backend:
  """what a statistical test needs from either side: a Tracer-like object for
  runSimulation / FreecadDocument.runSimulation and a one-call `hits(project, first, n, seed)`"""

  def __init__(self, name):
    self.name = name
    self._tracers = []

  def tracer(self, nthreads=4):
    if self.name == 'oracle':
      from oracle import capi
      from oracle_tracer import OracleTracer
      capi.build()
      t = OracleTracer(nthreads=nthreads)
    else:
      from freecad.optics_design_workbench_amd import _native
      from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
      _native.build()
      t = Tracer(0)
    self._tracers.append(t)
    return t

  def hits(self, pr, first, n, seed):
    """hit rows (HIT_DTYPE) of rays first..first+n-1 of the project's source"""
    if self.name == 'oracle':
      from oracle import capi
      capi.build()
      return capi.trace(pr.scene, pr.source, pr.limits, first, n, seed, nthreads=0)['hits']
    tr = self._tracers[0] if self._tracers else self.tracer()
    tr.setScene(pr.scene)
    tr.setSource(pr.source)
    tr.setLimits(pr.limits)
    tr.setDetector(None)
    tr.reserveHits(n * 2 + 1024)
    tr.reset()
    tr.trace(first, n, seed)
    tr.sync()
    assert tr.counters()['hits_dropped'] == 0
    return tr.hits()

  def traceRays(self, scene, lim, origins, dirs, wavelength=500.0):
    """hit rows of explicit rays, all groups recording"""
    import copy
    import numpy as np
    sc = copy.copy(scene)
    sc.group_record = np.ones_like(scene.group_record)
    if self.name == 'oracle':
      from oracle import capi
      capi.build()
      return capi.trace_rays(sc, lim, origins, dirs, wavelength=wavelength, nthreads=0)['hits']
    tr = self._tracers[0] if self._tracers else self.tracer()
    tr.setScene(sc)
    tr.setLimits(lim)
    tr.setDetector(None)
    tr.reserveHits(len(origins) * (lim.max_intersections + 1))
    tr.reset()
    tr.setWavelength(wavelength)
    tr.traceRays(origins, dirs)
    tr.sync()
    assert tr.counters()['hits_dropped'] == 0
    return tr.hits()

  def close(self):
    for t in self._tracers:
      t.close()


@pytest.fixture(params=BACKENDS)
def backend(request):
  b = _Backend(request.param)
  yield b
  b.close()
