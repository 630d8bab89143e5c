"""Scene-compiled kernels, the part that needs no GPU (odw_compile_check): the library writes the
header of constants of a baked scene and hiprtc compiles the ray loop against it for gfx950.
The launch side is tests/test_gpu_compiled.py."""
import re

import numpy as np
import pytest

from conftest import project

GRATING = 2      # ODW_OPT_GRATING (include/odw_trace.h)


def _table(header, name):
  m = re.search(r'constexpr \w[\w ]* ' + name + r'\(int i\) \{ constexpr [\w ]+ T\[\] = \{([^}]*)\}', header)
  assert m, name
  return [int(x.replace('ull', ''), 0) for x in m.group(1).split(',')]


@pytest.mark.parametrize('scene', ['minimal', 'lensesAndMirrors', 'lensesAndMirrorsSequential', 'GettingStarted',
                                   'grating', 'playground'])
def test_structure_header_and_compile(native_lib, scene):
  from freecad.optics_design_workbench_amd import _native
  proj = project(scene)
  sc = proj.scene
  header, code_bytes = _native.compile_check(sc, proj.limits, 'structure')
  assert code_bytes > 10000                       # a code object came out of hiprtc
  n = len(sc.prim_type)
  assert f'static constexpr int N = {n};' in header
  assert _table(header, 'type') == [int(t) for t in sc.prim_type]
  assert _table(header, 'group') == [int(g) for g in sc.prim_group]
  assert _table(header, 'gtype') == [int(t) for t in sc.group_type]
  assert _table(header, 'record') == [int(bool(r)) for r in sc.group_record]
  assert ('static constexpr bool seq() { return true; }' in header) == bool(sc.seq_enabled)
  # frame patterns: bit i = entry i is not zero; +-1 marks only where the entry is exactly +-1
  xf = _table(header, 'xf')
  m = np.asarray(sc.prim_xform).reshape(n, 12)
  for p in range(n):
    for i in range(12):
      assert bool(xf[p] >> i & 1) == (m[p, i] != 0.0)
      if i % 4 != 3:
        assert bool(xf[p] >> (12 + i) & 1) == (m[p, i] == 1.0)
        assert bool(xf[p] >> (24 + i) & 1) == (m[p, i] == -1.0)
  # no values in the header: the same text for another radius / position
  assert '0x1.' not in header and 'p+' not in header
  lean = not any(int(t) == GRATING for t in sc.group_type) and all(np.isinf(sc.group_abslen))
  assert f'#define ODW_SPEC_LEAN {"true" if lean else "false"}' in header


def test_structure_is_shared_by_a_parameter_sweep(native_lib):
  """the radius sweep of examples/1-getting-started changes values only: one header, one kernel"""
  import copy
  from freecad.optics_design_workbench_amd import _native, scenes
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  import os
  from conftest import SCENES
  doc = open_fcstd(os.path.join(SCENES, 'GettingStarted.FCStd'))
  headers = []
  for r in (9.0, 10.27, 11.0):
    doc.Sphere.Radius = r
    proj = scenes.bakeProject(doc)
    headers.append(_native.compile_check(proj.scene, proj.limits, 'structure')[0])
  assert headers[0] == headers[1] == headers[2]


@pytest.mark.parametrize('scene', ['hugeArray', 'edmund-optics-lens'])
def test_scenes_outside_the_flat_kernel_are_refused(native_lib, scene):
  """1500 spheres (grid kernel) and a lens with facets keep the generic kernels"""
  from freecad.optics_design_workbench_amd import _native
  proj = project(scene) if scene == 'hugeArray' else project(scene)
  if len(proj.scene.prim_type) <= 16 and not any(int(t) >= 5 for t in proj.scene.prim_type):
    pytest.skip('this bake of the scene fits the flat kernel')
  with pytest.raises(_native.NativeError, match='unsupported'):
    _native.compile_check(proj.scene, proj.limits, 'structure')


def test_frames_are_snapped_to_exact_zeros_and_ones():
  """rounding noise of the placement chain (cos 90 deg = 6e-17) would keep every frame product whole"""
  for scene in ('lensesAndMirrors', 'GettingStarted'):
    m = np.asarray(project(scene).scene.prim_xform).reshape(-1, 12)[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]]
    assert not ((np.abs(m) < 1e-12) & (m != 0)).any()
    assert not ((np.abs(np.abs(m) - 1) < 1e-12) & (np.abs(m) != 1)).any()
    r = m.reshape(-1, 3, 3)
    assert np.abs(r @ r.transpose(0, 2, 1) - np.eye(3)).max() < 1e-14


def test_random_structures_compile(native_lib):
  """random scenes of the parity fuzzers (boxes, spheres, cylinders, cones, tori; Common / Cut / Fuse; random
  optical types): every structure the flat kernel takes compiles for gfx950 (the launches are -m gpu:
  tests/test_gpu_fuzz.py, tests/fuzz_parity.py with ODW_COMPILE=structure)"""
  from freecad.optics_design_workbench_amd import _native
  from random_scenes import scene
  done = 0
  for s in range(40):
    rs = np.random.RandomState(4200 + s)
    try:
      sc, lim, targets = scene(rs, rich=(s % 3 == 2))
    except Exception:
      continue
    if len(sc.prim_type) > 16 or any(int(t) >= 5 for t in sc.prim_type):
      continue                                   # BVH / grid kernels
    header, code_bytes = _native.compile_check(sc, lim, 'structure')
    assert code_bytes > 10000
    assert f'static constexpr int N = {len(sc.prim_type)};' in header
    done += 1
    if done == 6:                                # (2 - 3 s per structure on one core)
      break
  assert done >= 6
