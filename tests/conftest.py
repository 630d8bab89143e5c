import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SCENES = os.path.join(GOLDEN, 'scenes')


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
