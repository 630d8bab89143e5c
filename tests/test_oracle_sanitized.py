"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build
only): every entry point on small inputs incl. the edge cases the GPU parity
tests use as reference (empty launches, zero capacity, trimmed solids, BVH-size
scene, stochastic surfaces, surface source).  Runs in a child process because
ASan must be the first library loaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ['ODW_ROOT']); sys.path.insert(0, os.path.join(os.environ['ODW_ROOT'], 'tests'))
import numpy as np
from oracle import capi
capi._LIB_PATH = os.path.join(os.environ['ODW_ROOT'], 'oracle', 'libodw_oracle_asan.so')
capi.build = lambda force=False: capi._LIB_PATH
from conftest import project
from freecad.optics_design_workbench_amd import scenes
for name, n in (('minimal', 300), ('lensesAndMirrors', 300), ('lensesAndMirrorsSequential', 200), ('hugeArray', 150),
                ('GettingStarted', 300), ('mirror-diffuse', 300), ('grating', 200), ('playground', 300)):
  pr = project(name)
  det = None
  if name in ('minimal', 'lensesAndMirrors'):
    det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=16, ny=16, toward=pr.source.xform[[3, 7, 11]])
  r = capi.trace(pr.scene, pr.source, pr.limits, 5, n, 3, det=det)
  assert r['counters']['traced_rays'] == n
  capi.trace(pr.scene, pr.source, pr.limits, 0, 0, 3)                       # empty launch
  r0 = capi.trace(pr.scene, pr.source, pr.limits, 0, 50, 3, hit_capacity=1)  # overflowing hit list
  assert r0['counters']['hits_dropped'] + len(r0['hits']) == r0['counters']['recorded_hits']
  t, phi = capi.sample(pr.source, 0, 64, 1)
  o, d = capi.make_rays(pr.source, 0, 64, 1)
  capi.trace_rays(pr.scene, pr.limits, o, d, surface_seed=2)
# tessellated shape (triangle primitives, interpolated normals) next to analytic ones
from freecad.optics_design_workbench_amd.freecad_elements import make, point_source
from freecad.optics_design_workbench_amd.scene import Document, bake
doc = Document()
make.makeLens(doc, [make.makeTessellated(doc, make.makeSphere(doc, 'S', 5, base=(0, 0, 30)), 12)], RefractiveIndex=1.5)
make.makeAbsorber(doc, [make.makeBox(doc, 'A', 100, 100, 1, base=(-50, -50, 60))])
make.makeSimulationSettings(doc)
src = make.makePointSource(doc, PowerDensity='exp(-theta**2/0.05**2)')
r = capi.trace(bake.bakeScene(doc, src), point_source.bakeSource(doc, src), bake.bakeLimits(doc, src), 0, 300, 1)
assert r['counters']['recorded_hits'] == 300
pr = project('simulation-modes-main')
o, d = capi.surface_rays(pr.source, 0, 500, 1)
capi.trace_surface(pr.scene, pr.source, pr.limits, 0, 500, 1)
assert capi.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
print('SANITIZED-OK')
'''


def test_oracle_under_asan_ubsan():
  so = os.path.join(ROOT, 'oracle', 'libodw_oracle_asan.so')
  subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libodw_oracle_asan.so'],
                        stdout=subprocess.DEVNULL)
  libasan = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], text=True).strip()
  if not os.path.isabs(libasan) or not os.path.exists(libasan):
    pytest.skip('libasan not available')
  env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1',
             UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1', ODW_ROOT=ROOT, OMP_NUM_THREADS='1')
  res = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0 and 'SANITIZED-OK' in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
