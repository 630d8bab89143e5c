"""Scene-compiled kernels on the device (odw_compile_scene): the same rays through the generic flat
kernel and through the kernel compiled against the scene -- same counters, same hit rows, same
histogram; cache behaviour; scenes outside the flat kernel's domain keep the generic kernels.
(The whole -m gpu suite also passes with ODW_COMPILE=structure, i.e. with every tracer compiling
its scenes: device-vs-oracle parity, randomised scenes and the acceptance tests included.)"""
import os

import numpy as np
import pytest

from conftest import SCENES, project

pytestmark = pytest.mark.gpu

SEED = 0x0D15EA5E


def run(tr, proj, n, det=None, first=0):
  tr.setScene(proj.scene)
  tr.setSource(proj.source)
  tr.setLimits(proj.limits)
  tr.setDetector(det)
  tr.reserveHits(max(16, 4 * n))
  tr.reset()
  tr.trace(first, n, SEED)
  tr.sync()
  return dict(counters=tr.counters(), hits=tr.hits(), hist=tr.histogram() if det is not None else None,
              info=tr.compiledInfo())


def same(a, b):
  assert a['counters'] == b['counters']
  assert np.array_equal(a['hits']['tag'], b['hits']['tag'])
  # the same arithmetic in the same order: the rows are equal bit for bit
  for col in ('point', 'direction', 'power'):
    assert np.array_equal(a['hits'][col], b['hits'][col]), col
  if a['hist'] is not None:
    assert np.array_equal(a['hist'], b['hist'])


@pytest.fixture()
def tracers(native_lib):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  made = []

  def make(mode):
    tr = Tracer(0)
    tr.compileScene(mode)
    made.append(tr)
    return tr
  yield make
  for tr in made:
    tr.close()


@pytest.mark.parametrize('scene,group', [('minimal', 'OpticalAbsorberGroup'), ('lensesAndMirrors', 'OpticalAbsorberGroup'),
                                         ('lensesAndMirrorsSequential', 'OpticalAbsorberGroup'), ('GettingStarted', None),
                                         ('grating', None), ('playground', None), ('mirror', None), ('mirror-diffuse', None)])
def test_compiled_kernel_equals_generic(tracers, scene, group, mode='structure'):
  from freecad.optics_design_workbench_amd import scenes
  proj = project(scene)
  det = None
  if group is not None:
    det = scenes.planeDetector(proj.scene, group, nx=128, ny=128, toward=proj.source.xform[[3, 7, 11]])
  n = 200000
  ref = run(tracers('off'), proj, n, det)
  got = run(tracers(mode), proj, n, det)
  assert ref['info']['mode'] == 0
  assert got['info']['mode'] == 1
  assert ref['counters']['traced_rays'] == n
  same(got, ref)


def test_all_groups_recording_and_explicit_rays(tracers):
  """every branch of the compiled interaction records; explicit initial conditions take the same kernel"""
  import copy
  proj = project('lensesAndMirrors')
  sc = copy.copy(proj.scene)
  sc.group_record = np.ones_like(sc.group_record)
  rng = np.random.default_rng(5)
  n = 100000
  o = np.tile(np.asarray(proj.source.xform, dtype=float).reshape(12)[[3, 7, 11]], (n, 1)) + rng.normal(0, 0.2, (n, 3))
  d = rng.normal(0, 0.02, (n, 3)) + np.array([0.0, 0.0, 1.0])
  out = []
  for mode in ('off', 'structure'):
    tr = tracers(mode)
    tr.setScene(sc)
    tr.setLimits(proj.limits)
    tr.setDetector(None)
    tr.reserveHits(16 * n)
    tr.reset()
    tr.traceRays(o, d)
    tr.sync()
    out.append(dict(counters=tr.counters(), hits=tr.hits(), hist=None))
  assert out[0]['counters']['recorded_hits'] > 2 * n
  same(out[1], out[0])


def test_one_kernel_for_a_parameter_sweep_and_process_cache(tracers):
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  doc = open_fcstd(os.path.join(SCENES, 'GettingStarted.FCStd'))
  tr, ref = tracers('structure'), tracers('off')
  infos = []
  for r in (9.0, 9.7, 10.27, 11.0):
    doc.Sphere.Radius = r
    proj = scenes.bakeProject(doc)
    got = run(tr, proj, 50000)
    same(got, run(ref, proj, 50000))
    infos.append(got['info'])
  assert all(i['mode'] == 1 for i in infos)
  # the first radius compiled (or found the kernel in a cache); the others reuse the loaded kernel
  assert all(i['cache'] == 1 and i['seconds'] == 0 for i in infos[1:])
  # another context of the process finds it too
  other = run(tracers('structure'), proj, 1000)
  assert other['info']['mode'] == 1 and other['info']['cache'] == 1


def test_scenes_outside_the_domain_keep_the_generic_kernels(tracers):
  proj = project('hugeArray')
  tr = tracers('structure')
  got = run(tr, proj, 20000)
  assert got['info']['mode'] == 0                 # grid kernel
  assert got['counters']['traced_rays'] == 20000
  # and back to a small scene: bound again
  small = run(tr, project('minimal'), 20000)
  assert small['info']['mode'] == 1


def test_histogram_window_edges(tracers):
  """the per-block LDS window of the histogram: histograms smaller than the window, a window clipped
  at the border, hits outside it -- always the bins of the plain count"""
  from freecad.optics_design_workbench_amd import scenes
  proj = project('lensesAndMirrors')
  n = 300000
  for nx, ny, half in ((16, 8, 10.0), (1024, 1024, 10.0), (2048, 64, 0.3), (200, 3000, 0.05)):
    det = scenes.planeDetector(proj.scene, 'OpticalAbsorberGroup', nx=nx, ny=ny, toward=proj.source.xform[[3, 7, 11]])
    det = dict(det, x_lo=-half, x_hi=half, y_lo=-half * 0.7, y_hi=half * 1.3)      # off-centre, partly cutting the spot
    got = run(tracers('structure'), proj, n, det)
    hist = got['hist'].reshape(nx, ny)
    # reference binning of the recorded rows on the host, with the kernel's arithmetic
    p = got['hits']['point'] - np.asarray(det['origin'])
    x, y = p @ np.asarray(det['ex']), p @ np.asarray(det['ey'])
    fx = np.floor((x - det['x_lo']) * (nx / (det['x_hi'] - det['x_lo'])))
    fy = np.floor((y - det['y_lo']) * (ny / (det['y_hi'] - det['y_lo'])))
    ok = (fx >= 0) & (fx < nx) & (fy >= 0) & (fy < ny)
    want = np.zeros((nx, ny), dtype=np.uint64)
    np.add.at(want, (fx[ok].astype(int), fy[ok].astype(int)), 1)
    assert got['counters']['hist_overflow'] == int((~ok).sum())
    assert np.array_equal(hist, want)


def test_a_kernel_that_cannot_be_built_leaves_the_generic_kernels(tracers, monkeypatch):
  """compiler trouble (here: a broken option) is reported by compileScene and does not stop the tracing"""
  from freecad.optics_design_workbench_amd import _native
  proj = project('lensesAndMirrors')
  ref = run(tracers('off'), proj, 50000)
  monkeypatch.setenv('ODW_SPEC_OPTS', '-DODW_SPEC_WAVES=oops')
  tr = tracers('off')
  tr.setScene(proj.scene)
  tr.setSource(proj.source)
  tr.setLimits(proj.limits)
  with pytest.raises(_native.NativeError, match='hiprtc'):
    tr.compileScene('structure')
  got = run(tr, proj, 50000)          # (setScene again: the bind fails again, silently this time)
  assert got['info']['mode'] == 0
  same(got, ref)


def test_small_paraboloid_scene_takes_the_compiled_kernel(tracers):
  """the generic flat kernel carries no paraboloid code (such documents go to the grid kernel); a compiled
  kernel has it exactly when the scene does: rays reflected off a parabolic mirror onto a screen, both routes"""
  from freecad.optics_design_workbench_amd.freecad_elements import make, point_source
  from freecad.optics_design_workbench_amd.scene import Document, bake
  import types
  doc = Document()
  f = 20.0
  pb = make.makeParaboloid(doc, 'P', f, 10.0, base=(0, 0, 50))
  make.makeMirror(doc, [pb], RecordHits=True)
  make.makeAbsorber(doc, [make.makeBox(doc, 'A', 200, 200, 1, base=(-100, -100, -30))], RecordHits=True)
  make.makeSimulationSettings(doc)
  src = make.makePointSource(doc, PowerDensity='exp(-theta**2/0.3**2)')
  proj = types.SimpleNamespace(scene=bake.bakeScene(doc, src), limits=bake.bakeLimits(doc, src),
                               source=point_source.bakeSource(doc, src))
  n = 100000
  ref = run(tracers('off'), proj, n)
  got = run(tracers('structure'), proj, n)
  assert ref['info']['mode'] == 0 and got['info']['mode'] == 1
  assert got['counters'] == ref['counters'] and ref['counters']['recorded_hits'] > n
  assert np.array_equal(got['hits']['tag'], ref['hits']['tag'])
  assert np.abs(got['hits']['point'] - ref['hits']['point']).max() < 1e-9
  assert np.abs(got['hits']['direction'] - ref['hits']['direction']).max() < 1e-9


def test_compiled_kernels_beyond_sixteen_primitives(tracers):
  """a train of eight lenses (25 primitives): generic flat kernel and compiled kernel, the same rows"""
  import types
  from freecad.optics_design_workbench_amd.freecad_elements import make, point_source
  from freecad.optics_design_workbench_amd.scene import Document, bake
  doc = Document()
  lenses = []
  for j in range(8):
    z = 30.0 + 12.0 * j
    a = make.makeSphere(doc, f'A{j}', 30.0, base=(0, 0, z + 28.0))
    b = make.makeSphere(doc, f'B{j}', 30.0, base=(0, 0, z - 28.0))
    c = make.makeCylinder(doc, f'C{j}', 6.0, 6.0, base=(0, 0, z - 3.0))
    lenses.append(make.makeCommon(doc, [a, b, c], f'L{j}'))
  make.makeLens(doc, lenses, RefractiveIndex=1.5)
  make.makeAbsorber(doc, [make.makeBox(doc, 'S', 60, 60, 1, base=(-30, -30, 30.0 + 12.0 * 8 + 20))], RecordHits=True)
  make.makeSimulationSettings(doc)
  src = make.makePointSource(doc, PowerDensity='exp(-theta**2/0.08**2)')
  proj = types.SimpleNamespace(scene=bake.bakeScene(doc, src), limits=bake.bakeLimits(doc, src),
                               source=point_source.bakeSource(doc, src))
  assert len(proj.scene.prim_type) == 25
  ref = run(tracers('off'), proj, 100000)
  got = run(tracers('structure'), proj, 100000)
  assert ref['info']['mode'] == 0 and got['info']['mode'] == 1
  assert ref['counters']['segments'] > 16 * 100000
  same(got, ref)


def test_auto_mode_compiles_in_the_background_and_switches_over(native_lib, monkeypatch, tmp_path):
  """ODW_COMPILE_AUTO: the first launches run the generic kernel, the scene earns its compilation by the rays it
  traces (ODW_SPEC_HOT_RAYS), a thread compiles, a later launch takes the compiled kernel -- rows identical
  throughout; with the kernel in a cache it is bound at once"""
  import time
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  monkeypatch.setenv('ODW_KERNEL_CACHE', str(tmp_path / 'kernels'))      # an empty disk cache: a real compilation
  monkeypatch.setenv('ODW_SPEC_HOT_RAYS', '150000')
  monkeypatch.setenv('ODW_SPEC_OPTS', '-DODW_TEST_AUTO_MODE=1')          # (a key no other test has loaded)
  proj = project('GettingStarted')
  n = 100000
  with Tracer(0) as ref_tr:
    ref = run(ref_tr, proj, n)
  with Tracer(0) as tr:
    assert tr.compileScene('auto')['mode'] == 0
    modes = []
    t0 = time.time()
    while time.time() - t0 < 60:
      got = run_again(tr, proj, n) if modes else run(tr, proj, n)
      same(got, ref)
      modes.append(got['info']['mode'])
      if modes[-1] == 2:
        break
      time.sleep(0.05)
    assert modes[0] == 0 and modes[1] == 0        # below the threshold, then compiling
    assert modes[-1] == 2, modes                  # switched over
    same(run_again(tr, proj, n), ref)
  with Tracer(0) as tr2:                          # the kernel is loaded: bound at once
    tr2.setScene(proj.scene); tr2.setSource(proj.source); tr2.setLimits(proj.limits)
    assert tr2.compileScene('auto')['mode'] == 2
    assert tr2.compiledInfo()['cache'] == 1


def run_again(tr, proj, n, first=0):
  """another launch of the same rays without uploading the scene again"""
  tr.reset()
  tr.trace(first, n, SEED)
  tr.sync()
  return dict(counters=tr.counters(), hits=tr.hits(), hist=None, info=tr.compiledInfo())


def test_many_short_runs_make_a_structure_hot(native_lib, monkeypatch, tmp_path):
  """the rays that earn a compilation are counted per scene structure over the whole process: a notebook's many
  short runs -- a new tracer and a new parameter value each -- get the compiled kernel like one long run"""
  import time
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  monkeypatch.setenv('ODW_KERNEL_CACHE', str(tmp_path / 'kernels'))
  monkeypatch.setenv('ODW_SPEC_HOT_RAYS', '100000')
  monkeypatch.setenv('ODW_SPEC_OPTS', '-DODW_TEST_SHORT_RUNS=1')
  doc = open_fcstd(os.path.join(SCENES, 'GettingStarted.FCStd'))
  modes = []
  t0 = time.time()
  k = 0
  while time.time() - t0 < 60 and (not modes or modes[-1] == 0):
    doc.Sphere.Radius = 9.0 + 0.01 * k
    k += 1
    proj = scenes.bakeProject(doc)
    with Tracer(0) as tr:
      tr.compileScene('auto')
      modes.append(run(tr, proj, 40000)['info']['mode'])
    time.sleep(0.05)
  assert modes[0] == 0 and modes[-1] == 2, modes
