"""Error behaviour of the C-ABI on a device: return codes instead of
exceptions at the boundary (include/odw_trace.h), converted to exceptions by
the binding like the reference raises ValueError / RuntimeError
(ray.py:235-237, random_number_generator.py:344-358, simulation_loop.py:715-723).
Edge cases: empty launches, overflowing hit lists, limits of the tables."""
import copy
import ctypes as C
import dataclasses

import numpy as np
import pytest

from conftest import project
from freecad.optics_design_workbench_amd import _native
from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import Document, bake

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tr(native_lib):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  t = Tracer(0)
  yield t
  t.close()


def test_calls_before_upload_fail_loudly(tr):
  with pytest.raises(_native.NativeError, match='no scene'):
    tr.trace(0, 10, 1)
  pr = project('minimal')
  tr.setScene(pr.scene)
  tr.setLimits(pr.limits)
  with pytest.raises(_native.NativeError, match='source'):
    tr.trace(0, 10, 1)                       # no source yet
  with pytest.raises(_native.NativeError):
    tr.sample(0, 10, 1)
  tr.setSource(pr.source)
  with pytest.raises(_native.NativeError, match='capacity'):
    tr.trace(0, 10, 1)                       # hit rows requested, no hit list reserved
  tr.trace(0, 10, 1, record_hits=False)      # fine without rows
  tr.sync()
  assert tr.counters()['traced_rays'] == 10
  with pytest.raises(ValueError):
    tr.histogram()                           # no detector set


def test_invalid_tables_are_rejected(tr, native_lib):
  pr = project('minimal')
  bad = copy.copy(pr.scene)
  bad.prim_group = pr.scene.prim_group + 7   # group index out of range
  with pytest.raises(_native.NativeError, match='invalid'):
    tr.setScene(bad)
  bad = copy.copy(pr.scene)
  bad.prim_type = np.full_like(pr.scene.prim_type, 9)
  with pytest.raises(_native.NativeError, match='unsupported'):
    tr.setScene(bad)
  src = copy.copy(pr.source)
  src.tables = copy.copy(pr.source.tables)
  src.tables.phi_cdf = pr.source.tables.phi_cdf * 0.5       # does not end at 1
  with pytest.raises(_native.NativeError, match='invalid'):
    tr.setSource(src)
  lim = copy.copy(pr.limits)
  lim.dist_tol = 0.0
  with pytest.raises(_native.NativeError):
    tr.setLimits(lim)
  with pytest.raises(_native.NativeError):
    tr.setDetector(dict(group=0, origin=[0, 0, 0], ex=[1, 0, 0], ey=[0, 1, 0], x_lo=1, x_hi=-1, y_lo=0, y_hi=1,
                        nx=4, ny=4))
  empty = _native.SceneDesc()
  empty.n_prims = 3                                           # counts without tables
  assert native_lib.odw_upload_scene(tr._ctx, C.byref(empty)) == 1
  assert b'null table' in native_lib.odw_last_error(tr._ctx)
  ctx = C.c_void_p()
  assert native_lib.odw_create(99, C.byref(ctx)) == 1         # bad device index
  assert native_lib.odw_create(0, None) == 1
  assert native_lib.odw_trace(None, 0, 1, 1, 0) == 1
  assert native_lib.odw_last_error(None)


def test_empty_and_overflowing_launches(tr, oracle):
  pr = project('lensesAndMirrors')
  tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits)
  tr.reserveHits(1000)
  tr.reset()
  tr.trace(0, 0, 1)                          # empty launch: nothing happens
  tr.sync()
  assert tr.counters()['traced_rays'] == 0 and len(tr.hits()) == 0
  tr.trace(0, 5000, 1)                       # more hits than capacity: dropped and counted, never written
  tr.sync()
  c = tr.counters()
  assert c['recorded_hits'] > 4900 and c['hits_dropped'] == c['recorded_hits'] - 1000
  h = tr.hits()
  assert len(h) == 1000
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 0, 5000, 1)
  assert c['recorded_hits'] == ref['counters']['recorded_hits']
  # the kept rows are rows of the full result (which ones is scheduling dependent)
  full = {(int(t), round(float(p[0]), 6)) for t, p in zip(ref['hits']['tag'], ref['hits']['point'])}
  assert all((int(t), round(float(p[0]), 6)) in full for t, p in zip(h['tag'], h['point']))
  tr.resetHits()
  assert tr.hitCount() == 0 and tr.counters()['recorded_hits'] == c['recorded_hits']   # counters keep running
  # a single ray, the last index of a 2^40 window
  tr.trace((1 << 40) - 1, 1, 1)
  tr.sync()
  assert tr.counters()['traced_rays'] == 5001


def test_group_and_sequence_limits(tr, oracle):
  """64 optical groups (ODW_MAX_GROUPS, the width of the relevance masks) and a
  100-step sequence (ODW_MAX_SEQUENCE); one more group is rejected on the host"""
  doc = Document()
  groups = []
  for i in range(64):
    kind = 'Vacuum' if i < 63 else 'Absorber'
    groups.append(make.makeOpticalGroup(doc, kind, [make.makeBox(doc, f'B{i}', 10, 10, 0.5, base=(-5, -5, 5 + 2 * i))],
                                        name=f'G{i}', RecordHits=True))
  seq = {f'SequentialModeElements_{i:02d}': [groups[i]] for i in range(64)}
  make.makeSimulationSettings(doc, SequentialMode=True, MaxIntersections=200.0, **seq)
  src = make.makePointSource(doc, PowerDensity='exp(-theta**2/1e-4)')
  from freecad.optics_design_workbench_amd.freecad_elements import point_source
  sc, lim, bs = bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)
  assert sc.n_groups == 64 and len(sc.seq_mask) == 64 and sc.n_prims == 64
  tr.setScene(sc); tr.setSource(bs); tr.setLimits(lim); tr.setDetector(None)
  n = 2000
  tr.reserveHits(n * 70)
  tr.reset()
  tr.trace(0, n, 3)
  tr.sync()
  g, gc = tr.hits(), tr.counters()
  ref = oracle.trace(sc, bs, lim, 0, n, 3, hit_capacity=n * 70)
  assert gc == ref['counters'] and gc['recorded_hits'] == 64 * n     # each step's plate once: entering only
  assert np.array_equal(g['tag'], ref['hits']['tag'])
  make.makeOpticalGroup(doc, 'Vacuum', [make.makeBox(doc, 'B64')], name='G64')
  with pytest.raises(ValueError):
    bake.bakeScene(doc, src)


def test_two_contexts_do_not_share_state(native_lib, oracle):
  """two contexts on one device with different scenes, sources, limits and
  detectors, launches interleaved without synchronising in between: each
  context's results equal its own oracle run"""
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  a, b = project('lensesAndMirrors'), project('hugeArray')
  det = scenes.planeDetector(a.scene, 'OpticalAbsorberGroup', nx=32, ny=32, toward=a.source.xform[[3, 7, 11]])
  with Tracer(0) as ta, Tracer(0) as tb:
    ta.setScene(a.scene); ta.setSource(a.source); ta.setLimits(a.limits); ta.setDetector(det)
    tb.setScene(b.scene); tb.setSource(b.source); tb.setLimits(b.limits); tb.setDetector(None)
    ta.reserveHits(60000); tb.reserveHits(60000)
    ta.reset(); tb.reset()
    for k in range(4):
      ta.trace(k * 10000, 10000, 5)
      tb.trace(k * 5000, 5000, 6)
    ta.sync(); tb.sync()
    ca, cb, ha = ta.counters(), tb.counters(), ta.histogram()
    rows_a = ta.hits()
  ra = oracle.trace(a.scene, a.source, a.limits, 0, 40000, 5, det=det, nthreads=8)
  rb = oracle.trace(b.scene, b.source, b.limits, 0, 20000, 6, nthreads=8)
  assert ca == ra['counters'] and np.array_equal(ha, ra['hist']) and np.array_equal(rows_a['tag'], ra['hits']['tag'])
  assert cb['traced_rays'] == 20000 and abs(cb['recorded_hits'] - rb['counters']['recorded_hits']) <= 0.01 * 20000


def _radius_variants(radii):
  from conftest import SCENES
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  import os
  out = []
  for r in radii:
    doc = open_fcstd(os.path.join(SCENES, 'GettingStarted.FCStd'))
    doc.Sphere.Radius = r
    out.append(scenes.bakeProject(doc))
  return out


def test_batch_entry_points_fail_loudly(tr, native_lib):
  """ABI v9 batch launches: what a sweep may not do, said as return codes + messages"""
  a, b = _radius_variants([9.8, 10.1])
  with pytest.raises(_native.NativeError, match='odw_set_limits'):
    tr.setSceneBatch([a.scene, b.scene])                       # the boxes carry the tolerance
  tr.setLimits(a.limits)
  with pytest.raises(_native.NativeError, match='odw_upload_scene_batch'):
    tr.traceBatch(0, 100, 1, 200)                              # no batch yet
  with pytest.raises(ValueError):
    tr.setSceneBatch([])
  # scenes of another structure, and scenes the flat kernels do not trace
  other = project('lensesAndMirrors')
  with pytest.raises(_native.NativeError, match='differs from scene 0 in structure'):
    tr.setSceneBatch([a.scene, other.scene])
  huge = project('hugeArray')
  tr.setLimits(huge.limits)
  with pytest.raises(_native.NativeError, match='flat kernels'):
    tr.setSceneBatch([huge.scene, huge.scene])
  with pytest.raises(_native.NativeError, match='odw_upload_scene_batch'):
    tr.traceBatch(0, 100, 1, 200)                              # the failed upload left no batch behind
  diffuse = project('mirror-diffuse')
  with pytest.raises(_native.NativeError, match='stochastic'):
    tr.setSceneBatch([diffuse.scene, diffuse.scene])
  assert native_lib.odw_upload_scene_batch(tr._ctx, None, 2) == 1
  assert native_lib.odw_upload_scene_batch(None, None, 2) == 1
  # a proper batch: rows need room, segments and counts have bounds
  tr.setLimits(a.limits)
  tr.setSceneBatch([a.scene, b.scene])
  with pytest.raises(_native.NativeError, match='source'):
    tr.traceBatch(0, 100, 1, 200)
  tr.setSource(a.source)
  with pytest.raises(_native.NativeError, match='rows_per_scene'):
    tr.traceBatch(0, 100, 1, 0)
  with pytest.raises(_native.NativeError, match='no such segment'):
    tr.batchSelect(0)                                          # nothing traced yet
  tr.traceBatch(0, 0, 1, 200)                                  # an empty launch is no launch
  tr.traceBatch(0, 1000, 1, 2000)
  tr.sync()
  rows, wanted = tr.batchRows()
  assert np.array_equal(rows, wanted) and np.all(rows > 900)
  with pytest.raises(_native.NativeError, match='no such segment'):
    tr.batchSelect(2)
  n3 = (C.c_uint64 * 3)()
  assert native_lib.odw_batch_rows(tr._ctx, n3, None, 3) == 1 and b'more scenes' in native_lib.odw_last_error(tr._ctx)
  assert native_lib.odw_batch_rows(tr._ctx, None, None, 1) == 1
  # a segment that is too small keeps what fits and says how much was asked for
  tr.traceBatch(0, 1000, 1, 100)
  tr.sync()
  rows, wanted = tr.batchRows()
  assert np.all(rows <= 100 + 64) and np.all(wanted > 900) and np.all(rows < wanted)
  tr.batchSelect(1)
  assert tr.hitCount() == rows[1]
  tr.batchSelect(None)
  # the segments belong to the last batch launch: one without rows leaves nothing to select
  tr.traceBatch(0, 1000, 1, 2000, record_hits=False)
  tr.sync()
  with pytest.raises(_native.NativeError, match='no such segment'):
    tr.batchSelect(0)
  with pytest.raises(_native.NativeError, match='no batch was traced with hit rows'):
    tr.batchRows()
  # a batch does not outlive the scene it was uploaded beside: a single-scene upload, or a tolerance that moves the
  # boxes, discards it (a rebuild of the one-scene tables under a batch launch would be read past its end)
  tr.setScene(a.scene)
  with pytest.raises(_native.NativeError, match='odw_upload_scene_batch'):
    tr.traceBatch(0, 1000, 1, 2000)
  tr.setSceneBatch([a.scene, b.scene])
  tr.traceBatch(0, 1000, 1, 2000)
  tr.sync()
  tr.setLimits(a.limits)                                        # the same tolerance: the batch stays
  tr.traceBatch(0, 1000, 1, 2000)
  tr.sync()
  looser = dataclasses.replace(a.limits, dist_tol=a.limits.dist_tol * 10)
  tr.setLimits(looser)
  with pytest.raises(_native.NativeError, match='odw_upload_scene_batch'):
    tr.traceBatch(0, 1000, 1, 2000)
  tr.setSceneBatch([a.scene, b.scene])
  tr.traceBatch(0, 1000, 1, 2000)
  tr.sync()
  rows, wanted = tr.batchRows()
  assert np.array_equal(rows, wanted) and np.all(rows > 900)


def test_archive_entry_points_fail_loudly(tr, native_lib):
  """ABI v9 rows kept in HBM"""
  with pytest.raises(_native.NativeError, match='nothing was archived'):
    tr.archiveSelect()
  tr.archiveSelect(False)                                      # back to the own list: always fine
  tr.archiveReset()
  assert native_lib.odw_archive_append(tr._ctx, None, None) == 1
  assert native_lib.odw_archive_select(None, 1) == 1 and native_lib.odw_archive_reset(None) == 1
  pr = project('minimal')
  tr.setScene(pr.scene); tr.setLimits(pr.limits); tr.setSource(pr.source)
  assert tr.archiveHits() == 0                                 # an empty list appends nothing
  with pytest.raises(_native.NativeError, match='nothing was archived'):
    tr.archiveSelect()
  tr.reserveHits(4000)
  tr.trace(0, 1000, 3)
  n = tr.archiveHits()
  tr.resetHits()
  tr.trace(1000, 1000, 3)
  assert tr.archiveHits() > n > 0
  tr.archiveSelect()
  both = tr.hits()
  tr.archiveSelect(False)
  own = tr.hits()
  ray = lambda h: (h['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
  assert ray(own).min() >= 1000 and ray(both).min() < 1000 and len(both) > len(own)
  tr.archiveReset()
  with pytest.raises(_native.NativeError, match='nothing was archived'):
    tr.archiveSelect()


def test_batched_post_hoc_steps_keep_their_order(tr, native_lib):
  """odw_batch_hits_*: select, then project, then bin -- of the last batch launch"""
  a, b = _radius_variants([9.9, 10.0])
  tr.setLimits(a.limits)
  tr.setSceneBatch([a.scene, b.scene])
  tr.setSource(a.source)
  pd, pu = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
  n, lv, od = (C.c_uint64 * 2)(), (C.c_uint64 * 2)(), (C.c_int32 * 2)()
  ex = np.tile([0.0, 1.0, 0.0], 2)
  ey = np.tile([0.0, 0.0, 1.0], 2)
  stats, mom = np.zeros(16), np.zeros(12)
  org, ea, eb = np.zeros(4), np.linspace(-1, 1, 5), np.linspace(-1, 1, 5)
  counts = np.zeros(2 * 16, dtype=np.uint64)
  lib = native_lib
  lib.odw_batch_hits_select.argtypes = [C.c_void_p, C.c_int32, pu, pu, C.POINTER(C.c_int32)]
  lib.odw_batch_hits_project.argtypes = [C.c_void_p, pd, pd, C.POINTER(C.c_int32), pd, pd]
  lib.odw_batch_hits_bin.argtypes = [C.c_void_p, C.c_int32, pd, pd, C.c_int32, pd, C.c_int32, pu]
  as_pd = lambda v: v.ctypes.data_as(pd)
  select = lambda: lib.odw_batch_hits_select(tr._ctx, -1, n, lv, od)
  project_ = lambda: lib.odw_batch_hits_project(tr._ctx, as_pd(ex), as_pd(ey), None, as_pd(stats), as_pd(mom))
  bin_ = lambda e=ea: lib.odw_batch_hits_bin(tr._ctx, 0, as_pd(org), as_pd(e), len(e), as_pd(eb), len(eb), counts.ctypes.data_as(pu))
  err = lambda: native_lib.odw_last_error(tr._ctx)
  assert select() == 1 and b'no batch was traced with hit rows' in err()
  assert project_() == 1 and b'odw_batch_hits_select first' in err()
  tr.traceBatch(0, 2000, 3, 4000)
  assert project_() == 1 and b'odw_batch_hits_select first' in err()
  assert select() == 0 and n[0] > 1900 and n[1] > 1900 and od[0] == 1
  assert bin_() == 1 and b'odw_batch_hits_project first' in err()
  assert project_() == 0
  for k in range(2):                      # the window around each scene's median spot
    org[2 * k], org[2 * k + 1] = stats[8 * k:8 * k + 2].mean(), stats[8 * k + 4:8 * k + 6].mean()
  assert bin_(np.array([0.0, 1.0, 0.5])) == 1 and b'monotonically' in err()
  assert lib.odw_batch_hits_bin(tr._ctx, 0, as_pd(org), as_pd(ea), 1, as_pd(eb), len(eb), counts.ctypes.data_as(pu)) == 1
  assert bin_() == 0 and counts.sum() > 0
  # a new launch, or a new batch, starts over
  tr.traceBatch(0, 2000, 4, 4000)
  assert bin_() == 1 and project_() == 1
  assert select() == 0
  tr.setSceneBatch([a.scene, b.scene])
  assert project_() == 1 and select() == 1


def test_enqueued_chain_keeps_its_order(tr, native_lib):
  """ABI v10, the chain that is enqueued and polled: begin -> sampled -> measure -> measured, each of the last batch
  launch; a piece asked for out of turn is refused with the name of the call that has to come first, polling a piece that
  is under way answers ODW_BUSY or the result, never an error, and the synchronous calls start over"""
  a, b = _radius_variants([9.9, 10.0])
  tr.setLimits(a.limits)
  tr.setSceneBatch([a.scene, b.scene])
  tr.setSource(a.source)
  lib, ctx = native_lib, tr._ctx
  pd, pu, pi = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
  lib.odw_batch_hits_begin.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
  lib.odw_batch_hits_sampled.argtypes = [C.c_void_p, C.c_int32, pu, pu, pi, C.c_void_p, C.c_uint64, pu]
  lib.odw_batch_hits_measure.argtypes = [C.c_void_p, pd, pd, pi, C.c_int32, pd, C.c_int32, pd, C.c_int32, C.c_uint64]
  lib.odw_batch_hits_measured.argtypes = [C.c_void_p, C.c_int32, pd, pd, pd, pu, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64, pu]
  n, lv, ns = (np.zeros(2, dtype=np.uint64) for _ in range(3))
  od = np.zeros(2, dtype=np.int32)
  cap = 308
  rows = np.zeros((2, cap), dtype=_native.HIT_DTYPE)
  ex, ey = np.tile([0.0, 1.0, 0.0], 2), np.tile([0.0, 0.0, 1.0], 2)
  ea, eb = np.linspace(-30, 30, 7), np.linspace(-30, 30, 7)
  stats, mom, org = np.zeros(16), np.zeros(12), np.zeros(4)
  counts, flags, nk = np.zeros(2 * 36, dtype=np.uint64), np.zeros(2, dtype=np.uint32), np.zeros(2, dtype=np.uint64)
  as_ = lambda v, t: v.ctypes.data_as(t)
  begin = lambda limit=300: lib.odw_batch_hits_begin(ctx, -1, limit)
  sampled = lambda wait, c=cap: lib.odw_batch_hits_sampled(ctx, wait, as_(n, pu), as_(lv, pu), as_(od, pi), rows.ctypes.data_as(C.c_void_p), c, as_(ns, pu))
  measure = lambda e=ea: lib.odw_batch_hits_measure(ctx, as_(ex, pd), as_(ey, pd), None, 0, as_(e, pd), len(e), as_(eb, pd), len(eb), 0)
  measured = lambda wait: lib.odw_batch_hits_measured(ctx, wait, as_(stats, pd), as_(mom, pd), as_(org, pd), as_(counts, pu),
                                                      flags.ctypes.data_as(C.POINTER(C.c_uint32)), None, 0, as_(nk, pu))
  err = lambda: lib.odw_last_error(ctx)
  assert begin() == 1 and b'no batch was traced with hit rows' in err()
  assert sampled(1) == 1 and b'odw_batch_hits_begin first' in err()
  tr.traceBatch(0, 3000, 3, 6000)
  assert begin(0) == 1 and begin(5000) == 1                     # the sample's size has bounds
  assert measure() == 1 and b'odw_batch_hits_sampled first' in err()
  assert measured(1) == 1 and b'odw_batch_hits_measure first' in err()
  assert begin() == 0
  assert sampled(1, 10) == 4 and b'too small' in err()          # ODW_ERR_CAPACITY; the piece is still there
  while True:
    rc = sampled(0)
    assert rc in (0, _native.BUSY), err()
    if rc == 0:
      break
  assert n.min() > 2800 and od.tolist() == [1, 1] and 0 < ns.min() <= 300
  assert sampled(1) == 1                                        # handed over once
  assert measured(0) == 1 and b'odw_batch_hits_measure first' in err()
  assert measure(np.array([0.0, 1.0, 0.5])) == 1 and b'monotonically' in err()
  assert measure() == 0
  assert measure() == 1                                         # one measure per sample
  while True:
    rc = measured(0)
    assert rc in (0, _native.BUSY), err()
    if rc == 0:
      break
  assert flags.tolist() == [0, 0] and counts.reshape(2, -1).sum(axis=1).min() > 2800
  assert measured(1) == 1
  # a new launch starts over; so does a chain begun again
  tr.traceBatch(0, 3000, 4, 6000)
  assert sampled(1) == 1 and measure() == 1
  assert begin() == 0 and begin() == 0 and sampled(1) == 0
