"""Post-hoc detector binning on the rows in HBM (`DeviceHits`, csrc/odw_posthoc.hip) against
(1) outputs of the reference's own Hits / Histogram classes (tests/golden/hist_cases.npz,
fwhm_cases.npz) and (2) the host `Hits` on the rows of a real trace: same plane, same origin,
same edges, same counts."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, project

pytestmark = pytest.mark.gpu
KINDS = {'cart30': dict(bins=30), 'polar3x50': dict(bins=(3, 50), binCoords='polar'),
         'cartlin': dict(bins=[np.linspace(-2, 2, 41), np.linspace(-1, 1, 21)])}


@pytest.fixture(scope='module')
def tracer(native_lib):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  tr = Tracer(0)
  yield tr
  tr.close()


def _same(H, G):
  assert np.array_equal(H._planeNormal, G._planeNormal) and np.array_equal(H._xInPlaneVec, G._xInPlaneVec)
  assert np.array_equal(H._origin, G._origin)
  assert np.array_equal(H.binX, G.binX) and np.array_equal(H.binY, G.binY)
  assert np.array_equal(H.hist, G.hist)
  assert np.array_equal(np.asarray(H.binAreas), np.asarray(G.binAreas))


@pytest.mark.parametrize('tag', ['A', 'B'])
@pytest.mark.parametrize('kind', list(KINDS))
def test_device_histogram_equals_reference_classes(tracer, tag, kind):
  g = np.load(os.path.join(GOLDEN, 'hist_cases.npz'))
  P, D = g[f'{tag}_points'], g[f'{tag}_directions']
  dh = tracer.loadHits(dict(points=P, directions=D, powers=np.ones(len(P)), isEntering=np.ones(len(P), dtype=int)))
  assert len(dh) == len(P)
  H = dh.histogram(**KINDS[kind])
  key = f'{tag}_{kind}'
  assert np.array_equal(H._planeNormal, g[key + '_normal'])
  assert np.array_equal(H._xInPlaneVec, g[key + '_xvec'])
  # the projection is three products summed left to right here, a BLAS dot in numpy: the median can
  # differ in its last bits, counts only if a hit sits within that of an edge
  assert np.abs(H._origin - g[key + '_origin']).max() < 1e-13
  assert np.allclose(H.binX, g[key + '_binX'], rtol=0, atol=1e-12) and np.allclose(H.binY, g[key + '_binY'], rtol=0, atol=1e-12)
  assert np.array_equal(H.hist, g[key + '_hist'])
  if kind.startswith('polar'):
    assert np.allclose(H.byAzimuth()[2], g[key + '_az_dens'], rtol=1e-12)


@pytest.mark.parametrize('tag', ['tight', 'wide', 'sparse'])
def test_device_calc_fwhm_equals_reference(tracer, tag):
  from freecad.optics_design_workbench_amd.simulation import sweep
  g = np.load(os.path.join(GOLDEN, 'fwhm_cases.npz'))
  P, D = g[tag + '_points'], g[tag + '_directions']
  dh = tracer.loadHits(dict(points=P, directions=D, powers=np.ones(len(P)), isEntering=np.ones(len(P), dtype=int)))
  H = dh.histogram(binCoords='polar', bins=[np.arange(0, 2 * np.pi, np.pi / 2), np.geomspace(1e-3, 5, 500)])
  assert np.array_equal(H.hist, g[tag + '_hist'])
  assert sweep.calcFwhm(dh) == pytest.approx(float(g[tag + '_fwhm']), rel=1e-12)


def test_device_histogram_equals_host_hits_on_traced_rows(tracer):
  """rows of a real launch (lensesAndMirrors, 1e6 rays; and GettingStarted with leaving rows of a
  transparent detector mixed in): DeviceHits on the rows in HBM == Hits on the fetched rows"""
  from freecad.optics_design_workbench_amd.jupyter_utils import Hits
  from freecad.optics_design_workbench_amd.simulation import sweep
  from freecad.optics_design_workbench_amd.simulation.tracer import hitsToDict
  import copy
  for name, record_all in (('lensesAndMirrors', False), ('GettingStarted', False), ('GettingStarted', True)):
    pr = project(name)
    sc = pr.scene
    if record_all:
      sc = copy.copy(sc)
      sc.group_record = np.ones_like(sc.group_record)       # lens rows too: entering and leaving hits
    n = 1_000_000
    tracer.setScene(sc)
    tracer.setSource(pr.source)
    tracer.setLimits(pr.limits)
    tracer.setDetector(None)
    tracer.reserveHits(4 * n)
    tracer.reset()
    tracer.trace(0, n, 5)
    tracer.sync()
    rows = tracer.hits()
    for group in ([None] if not record_all else [None, sc.group_index('OpticalLensGroup')]):
      sel = rows if group is None else rows[((rows['tag'] >> np.uint64(48)) & np.uint64(0x7FFF)) == group]
      host = Hits(dict(points=np.ascontiguousarray(sel['point']), directions=np.ascontiguousarray(sel['direction']),
                       powers=np.ascontiguousarray(sel['power']), isEntering=(sel['tag'] >> np.uint64(63)).astype(np.int64)))
      dh = tracer.deviceHits(group)
      assert len(dh) == len(sel) > 0
      pn, xv = dh.detectPlaneNormal()
      hn, hx = host.detectPlaneNormal()
      assert np.array_equal(pn, hn) and np.array_equal(xv, hx)
      for kw in (dict(bins=30), dict(bins=(3, 50), binCoords='polar'), dict(bins=[40, np.linspace(-1, 1, 33)]),
                 dict(radius=0.5, bins=64), dict(binCoords='polar', radius=2.0, bins=(4, 100)),
                 dict(binCoords='polar', bins=[np.arange(0, 2 * np.pi, np.pi / 2), np.geomspace(1e-3, 5, 500)])):
        H, G = dh.histogram(**kw), host.histogram(**kw)
        assert np.abs(H._origin - G._origin).max() < 1e-12
        assert np.allclose(H.binX, G.binX, rtol=0, atol=1e-11) and np.allclose(H.binY, G.binY, rtol=0, atol=1e-11)
        assert H.hist.sum() == G.hist.sum()
        assert np.abs(H.hist - G.hist).sum() <= 2             # a hit within an ulp of an edge may change sides
      assert dh.rmsSpot() == pytest.approx(sweep.rmsSpot(host), rel=1e-9)
      assert sweep.calcFwhm(dh) == pytest.approx(sweep.calcFwhm(host), rel=1e-6, nan_ok=True)
      # (the moment sums above rode along with the projection of the points; a new selection adds them up by
      #  themselves: the same bits)
      cached = dh.moments()
      dh = tracer.deviceHits(group)
      alone = dh.moments()
      assert np.array_equal(cached[0], alone[0]) and np.array_equal(cached[1], alone[1])
      tracer.hits()                                          # a fetch ends the selection ...
      with pytest.raises(Exception):
        dh.histogram(bins=10)                                # ... and the object says so


def test_streaming_fetch_yields_every_row_of_every_job(tracer):
  """two hit lists, the copy of job k overlapping the trace of job k+1: every job's rows arrive,
  complete and uncorrupted (same multiset as the ordered fetch of the same job)"""
  pr = project('lensesAndMirrors')
  tracer.setScene(pr.scene)
  tracer.setSource(pr.source)
  tracer.setLimits(pr.limits)
  tracer.setDetector(None)
  n = 3_000_000
  jobs = [(k * n, n) for k in range(5)]
  tracer.reset()
  got = [np.sort(chunk['tag'].copy()) for chunk in tracer.traceStreaming(iter(jobs), 9, capacity=n + 1024, histogram=False)]
  assert len(got) == len(jobs)
  assert tracer.counters()['traced_rays'] == n * len(jobs) and tracer.counters()['hits_dropped'] == 0
  for (first, m), tags in zip(jobs, got):
    tracer.reset()
    tracer.trace(first, m, 9)
    tracer.sync()
    want = tracer.hits()
    assert np.array_equal(tags, np.sort(want['tag']))


def test_hit_columns_equal_the_split_rows(native_lib):
  """Tracer.hitColumns (odw_hits_select + odw_hits_columns): per recording group the arrays of the reference's
  hit dictionary, gathered on the device = the fetched rows split by group on the host"""
  import copy
  from conftest import project
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  proj = project('lensesAndMirrors')
  sc = copy.copy(proj.scene)
  sc.group_record = np.array([1, 0, 1, 1], dtype=np.int32)       # two mirrors' groups and the absorber
  n = 200000
  with Tracer(0) as tr:
    tr.setScene(sc); tr.setSource(proj.source); tr.setLimits(proj.limits); tr.setDetector(None)
    tr.reserveHits(4 * n)
    tr.trace(5, n, 77)
    tr.sync()
    rows = tr.hits()
    grp = ((rows['tag'] >> np.uint64(48)) & np.uint64(0x7FFF)).astype(int)
    seen = 0
    for g in range(4):
      cols = tr.hitColumns(g)
      sel = rows[grp == g]
      if len(sel) == 0:
        assert cols is None
        continue
      seen += 1
      assert np.array_equal(cols['points'], sel['point']) and np.array_equal(cols['directions'], sel['direction'])
      assert np.array_equal(cols['powers'], sel['power'])
      assert np.array_equal(cols['isEntering'], (sel['tag'] >> np.uint64(63)).astype(np.int64))
      assert np.array_equal(cols['rayIndex'], (sel['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64))
    assert seen == 3


def test_selection_without_a_sort_is_the_sorted_selection(native_lib):
  """`odw_hits_select` orders a group's rows by ray index: by rank in a bitmap of rays where no ray has two selected
  rows (an absorbing detector: GettingStarted), by radix sort otherwise (every group recording: several rows per
  ray) -- ODW_SELECT_SORT=1 forces the sort (read once per process: a child process per route).  Both routes: the same
  columns, row for row."""
  import subprocess
  import sys
  import tempfile
  here = os.path.dirname(os.path.abspath(__file__))
  n = 300000
  code = f"""
import sys, copy, numpy as np
sys.path.insert(0, {os.path.dirname(here)!r}); sys.path.insert(0, {here!r})
from conftest import project
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
pr = project('GettingStarted')
every, group = sys.argv[2] == '1', sys.argv[3]
sc = copy.copy(pr.scene)
if every:
  sc.group_record = np.ones_like(pr.scene.group_record)
with Tracer(0) as tr:
  tr.setScene(sc); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
  tr.reserveHits({n} * 8); tr.reset(); tr.trace(1000, {n}, 5, histogram=False); tr.sync()
  c = tr.hitColumns(-1 if group == 'all' else sc.group_index(group), pinned=False)
  np.savez(sys.argv[1], **{{k: np.asarray(v) for k, v in c.items()}})
"""
  for every, group, unique in (('0', 'OpticalAbsorberGroup', True), ('1', 'all', False), ('1', 'OpticalLensGroup', False)):
    cols = {}
    for forced in (False, True):
      env = {k: v for k, v in os.environ.items() if k != 'ODW_SELECT_SORT'}
      if forced:
        env['ODW_SELECT_SORT'] = '1'
      with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'cols.npz')
        res = subprocess.run([sys.executable, '-c', code, out, every, group], capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        cols[forced] = dict(np.load(out))
    a, b = cols[False], cols[True]
    assert set(a) == set(b) and len(a['rayIndex']) > 0.5 * n
    for k in a:
      assert np.array_equal(a[k], b[k]), (group, k)
    steps = np.diff(a['rayIndex'])
    assert np.all(steps >= 0) and bool(np.all(steps > 0)) == unique


def test_a_run_keeps_its_rows_on_the_device(native_lib, tmp_path):
  """runSimulation (keepOnDevice 'auto', the default, and True): the rows of every launch join an archive in HBM
  (odw_archive_append: device to device, from the fetch threads of the overlapped loop onto a context of its own);
  RawFolder.loadHits() of the same process is lazy about the files and bins in HBM; loadHits(device=True)
  bins them where they are -- the Histogram of the files read back (plane, origin, counts), the same moments, the same
  rows; a later run releases them; without the switch loadHits(device=True) reads the files"""
  import shutil
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument
  from freecad.optics_design_workbench_amd.simulation import results_store
  from freecad.optics_design_workbench_amd.simulation.device_hits import DeviceHits
  path = str(tmp_path / 'GettingStarted.FCStd')
  shutil.copy(os.path.join(SCENES, 'GettingStarted.FCStd'), path)
  with FreecadDocument(path) as f:
    f.OpticalSimulationSettings.EndAfterRays = '3e5'
    for overlap in (True, False):
      # (keepOnDevice is the default since round 5: 'auto')
      raw = f.runSimulation('true', raysPerLaunch=70000, overlapFetch=overlap, **({} if overlap else dict(keepOnDevice=True)))
      host = raw.loadHits(device=False)
      assert type(host) is results_store.Hits
      dev = raw.loadHits(device=True)
      assert isinstance(dev, DeviceHits) and len(dev) == len(host) > 2.9e5
      kw = dict(binCoords='polar', bins=[np.arange(0, 2 * np.pi, np.pi / 2), np.geomspace(1e-3, 5, 200)])
      a, b = dev.histogram(**kw), host.histogram(**kw)
      assert np.array_equal(a.hist, b.hist) and np.allclose(a._origin, b._origin, atol=1e-12)
      assert np.array_equal(a._planeNormal, b._planeNormal)
      # what a notebook gets without asking for anything: the files are not read until an array is wanted, histograms
      # and the plane come from the rows in HBM
      lazy = raw.loadHits()
      assert isinstance(lazy, results_store.RunHits) and lazy._loaded is None
      c = lazy.histogram(**kw)
      assert len(lazy) == len(host) and lazy._loaded is None
      assert np.array_equal(c.hist, b.hist) and np.array_equal(c._planeNormal, b._planeNormal) and np.allclose(c._origin, b._origin, atol=1e-12)
      n1, x1 = lazy.detectPlaneNormal()
      n2, x2 = host.detectPlaneNormal()
      assert np.array_equal(n1, n2) and np.array_equal(x1, x2) and lazy._loaded is None
      assert np.array_equal(lazy.points(), host.points()) and lazy._loaded is not None      # (now the arrays are there ...)
      d = lazy.histogram(**kw)                                                               # (... and numpy bins them)
      assert np.array_equal(d.hist, b.hist) and np.array_equal(d._origin, b._origin)
      rows = dev.toHits()
      order = np.lexsort(host.points().T[::-1])
      assert np.array_equal(rows.points()[np.lexsort(rows.points().T[::-1])], host.points()[order])
    assert len(results_store._DEVICE_RUNS) == 1                # (the second run released the first one's rows)
    plain = f.runSimulation('true', raysPerLaunch=70000, keepOnDevice=False)
    assert not isinstance(plain.loadHits(device=True), DeviceHits) and type(plain.loadHits()) is results_store.Hits
  results_store.releaseDeviceRuns()
  assert not results_store._DEVICE_RUNS


def test_polar_binning_decides_every_row_like_numpy(tracer):
  """polar bins row by row (ph_bin_kernel): rows a hair from an azimuth edge, on it, with signed zeros, subnormal and
  tiny coordinates, and a random bulk -- the counts are numpy.histogram2d's on arctan2 / sqrt of the same coordinates.
  (Written for a variant that screened the azimuth in float32 and skipped rows with a negative first coordinate; it
  counted the same and ran slower -- 148 against 119 us per 1e7 rows, profiles/r04/README.md -- and was dropped; the
  test stays for whatever comes next.)"""
  rng = np.random.default_rng(11)
  xs, ys = [], []
  for edge in (0.0, np.pi / 2, np.pi, -np.pi / 2, -np.pi, 1.0, -2.5):
    # (not +-4e-16: at pi - 3.2e-16 ocml's arctan2 and libm's round to different neighbours -- one ulp, the plain kernel
    #  and the tables alike; "a hit within an ulp of an edge may change sides")
    for delta in (0.0, 1e-16, -1e-16, 1.5e-15, -1.5e-15, 2.5e-15, -2.5e-15, 1e-14, -1e-14, 1e-13, -1e-13, 1e-10, -1e-10,
                  1e-8, -1e-8, 9e-6, -9e-6, 1.1e-5, -1.1e-5, 1e-4, -1e-4):
      for r in (1.5e-3, 0.1, 3.0, 700.0):
        a = edge + delta
        xs.append(r * np.sin(a))
        ys.append(r * np.cos(a))
  special = [(0.0, 1.0), (0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (-0.0, 1.0), (-0.0, -1.0), (-1e-300, 1e-3), (-5e-324, 1000.0),
             (5e-324, 1000.0), (1e-40, 1e-40), (1e-35, -1e-35), (-1e-33, 2e-33), (2e-3, 1e-300), (-2e-3, -1e-300),
             (1e-290, 1.0), (-1e-290, 1.0), (-1e-270, 1.0)]
  xs += [s[0] for s in special]
  ys += [s[1] for s in special]
  bulk = rng.normal(size=(200_000, 2)) * np.array([0.3, 0.3])
  radial = np.geomspace(1e-3, 2000.0, 300)
  # rows whose radius is a radial edge, or the double next to one (the table over the bit patterns of the radius,
  # round 5: PhbBinAccel), along directions whose squares add up without rounding
  for e in np.r_[radial[:5], radial[148:152], radial[-3:], np.linspace(0.5, 40.0, 12)]:
    for v in (e, np.nextafter(e, 0.0), np.nextafter(e, np.inf)):
      xs += [v, 0.0, 0.6 * v]
      ys += [0.0, v, 0.8 * v]
  X = np.r_[xs, bulk[:, 0]]
  Y = np.r_[ys, bulk[:, 1]]
  # the plane z = 0 seen against +z: x in the plane = (1, 0, 0), y = n x x = (0, -1, 0); with these axes the
  # projection is exact, so the device bins exactly (X, Y)
  P = np.c_[X, -Y, np.zeros(len(X))]
  D = np.tile([0.0, 0.0, 1.0], (len(X), 1))
  dh = tracer.loadHits(dict(points=P, directions=D, powers=np.ones(len(P)), isEntering=np.ones(len(P), dtype=int)))
  import os
  x, y = X + 0.0 - 0.0, Y + 0.0 - 0.0      # (a projection sums three products: -0.0 + 0.0 = +0.0)
  for azimuth in (np.arange(0, 2 * np.pi, np.pi / 2), np.linspace(-np.pi, np.pi, 9), np.linspace(-np.pi, np.pi, 8), np.array([-3.0, -2.5, 1.0, 3.0]),
                  np.array([0.5, 1.0, 2.0]), np.array([-4.0, -np.pi, 0.0, np.pi, 4.0])):
    for rad in (radial, np.linspace(0.5, 40.0, 12), np.array([1e-3, 1e-3, 0.2, 0.2, 7.0])):
      want, _, _ = np.histogram2d(np.arctan2(x, y), np.sqrt(x * x + y * y), bins=[azimuth, rad])
      # (with the tables of round 5 -- cross products for the azimuth, a guide over the radius' bit pattern --, and plain)
      for plain in (False, True):
        os.environ.pop('ODW_BIN_PLAIN', None)
        if plain:
          os.environ['ODW_BIN_PLAIN'] = '1'
        try:
          H = dh.histogram(planeNormal=np.array([0.0, 0.0, -1.0]), xInPlaneVec=np.array([1.0, 0.0, 0.0]), origin=np.zeros(2),
                           binCoords='polar', bins=[azimuth, rad])
        finally:
          os.environ.pop('ODW_BIN_PLAIN', None)
        assert np.array_equal(H._planeNormal, [0.0, 0.0, -1.0])
        assert np.array_equal(H.hist, want), (plain, len(azimuth), len(rad), np.argwhere(H.hist != want)[:5])
    assert want.sum() > 1000
