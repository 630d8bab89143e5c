"""Run-folder contract: files this package writes follow the
reference's layout and dictionary keys (results_store.py:352-367, 405-457) and
load back through RawFolder the way freecad_document.py:1485-1504 does."""
import os
import pickle

import numpy as np
import pytest


def _fill(store, n=100, seed=0):
  rs = np.random.RandomState(seed)
  store.addRayHits('OpticalPointSource', 'src', 'OpticalAbsorberGroup', 'OpticalAbsorberGroup',
                   rs.rand(n, 3), rs.rand(n, 3), np.ones(n), np.ones(n, dtype=int))
  store.incrementRayCount(n)
  store.incrementIterationCount(1)


def test_run_folder_layout_and_round_trip(tmp_path):
  from freecad.optics_design_workbench_amd.simulation import results_store as rs
  res = rs.resultsFolderPath(str(tmp_path / 'proj.FCStd'))
  assert res.endswith('proj.OpticsDesign')
  store = rs.SimulationResults('true', resultsPath=res)
  assert store.simulationRunFolder == 'raw/simulation-run-000000'
  _fill(store, 100, 0)
  store.flush()
  _fill(store, 50, 1)
  store.flush()
  run = store.runFolderPath()
  assert any(f.startswith('uid-') for f in os.listdir(run))
  folder = os.path.join(run, 'source-src', 'object-OpticalAbsorberGroup')
  files = sorted(os.listdir(folder))
  assert len(files) == 2 and all(f.endswith('-hits.pkl') and '-pid' in f and '-thread' in f for f in files)
  d = pickle.load(open(os.path.join(folder, files[0]), 'rb'))
  assert set(d) == {'source', 'obj', 'points', 'directions', 'powers', 'isEntering'}
  assert d['source'] == 'OpticalPointSource' and d['obj'] == 'OpticalAbsorberGroup'
  assert d['points'].shape == (100, 3) and d['isEntering'].dtype.kind == 'i'
  raw = rs.latestRawFolder(res)
  h = raw.loadHits('*')
  assert len(h) == 150 and h.hits['points'].shape == (150, 3)
  assert np.array_equal(h.hits['points'], store.hits().hits['points'])
  # a second run gets the next index
  store2 = rs.SimulationResults('true', resultsPath=res)
  assert store2.simulationRunFolder == 'raw/simulation-run-000001'
  assert len(rs.rawFolders(res)) == 2


def test_end_criteria_are_strict():
  from freecad.optics_design_workbench_amd.simulation import results_store as rs
  s = rs.SimulationResults('true', endAfterRays=100)
  s.incrementRayCount(100)
  assert not s.reachedEnd()
  s.incrementRayCount(1)
  assert s.reachedEnd()
  s = rs.SimulationResults('true', endAfterHits=10)
  _fill(s, 11)
  assert s.reachedEnd()


def test_update_result_entry():
  from freecad.optics_design_workbench_amd.simulation.results_store import updateResultEntry
  r = {}
  updateResultEntry(r, 'source', 'a')
  updateResultEntry(r, 'source', 'a')
  updateResultEntry(r, 'points', np.zeros((2, 3)))
  updateResultEntry(r, 'points', np.ones((3, 3)))
  assert r['source'] == 'a' and r['points'].shape == (5, 3)


@pytest.mark.gpu
def test_run_simulation_end_criteria_on_device(native_lib, tmp_path):
  """test/21-simulation-modes/run-simulations.py:47-69 restated: EndAfterHits=1e3
  => len(hits) > 999; EndAfterRays=1e3 => > 100 hits; endIf callback"""
  import shutil
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from freecad.optics_design_workbench_amd.simulation import runSimulation, resultsFolderPath, latestRawFolder
  path = str(tmp_path / 'GettingStarted.FCStd')
  shutil.copy(os.path.join(SCENES, 'GettingStarted.FCStd'), path)
  doc = open_fcstd(path)
  st = doc.OpticalSimulationSettings
  st.EndAfterRays, st.EndAfterHits = 'inf', '1e3'
  store = runSimulation(doc, 'true', resultsPath=resultsFolderPath(path), raysPerLaunch=1 << 16)
  assert len(store.hits()) > 999
  raw = latestRawFolder(resultsFolderPath(path))
  assert len(raw.loadHits('*')) == len(store.hits())
  # run-folder extras: global info, master progress summaries, status flags
  assert raw.loadGlobalInfo()['opticalObjects'][0]['placementPathsAndMatrices'][0]['gpM'].shape == (4, 4)
  assert raw.loadProgress()['totalRecordedHits'] == len(store.hits())
  flags = set(os.listdir(resultsFolderPath(path)))
  assert 'simulation-is-done' in flags and not flags & {'simulation-is-running', 'simulation-is-canceled'}
  st.EndAfterRays, st.EndAfterHits = '1e3', 'inf'
  store = runSimulation(doc, 'true')
  assert store.totalTracedRays > 1000 and store.totalTracedRays <= 1100 and len(store.hits()) > 100
  st.EndAfterRays = 'inf'
  store = runSimulation(doc, 'true', endIf=lambda s: s.totalRecordedHits > 1e3, raysPerLaunch=1 << 12)
  assert len(store.hits()) > 1e3
  with pytest.raises(ValueError):
    runSimulation(doc, 'true')            # no end criterion at all
  # single-shot modes
  store = runSimulation(doc, 'singletrue')
  assert store.totalTracedRays == 100 and store.totalIterations == 1
  store = runSimulation(doc, 'fans')
  # metadata travels with the hits only if the settings' StoreHit* switches are on (ray.py:55-65)
  assert store.totalTracedRays == 40 and set(store.hits().hits) == {'source', 'obj', 'points', 'directions',
                                                                    'powers', 'isEntering'}
  assert not store.hits().supportsFanMath()
  for k in ('StoreHitFanIndex', 'StoreHitRayIndex', 'StoreHitTotalRaysInFan', 'StoreHitTotalFanCount',
            'StoreHitInitPhi', 'StoreHitInitTheta'):
    setattr(st, k, True)
  store = runSimulation(doc, 'fans')
  h = store.hits().hits
  assert {'fanIndex', 'rayIndex', 'totalRaysInFan', 'totalFanCount', 'initPhi', 'initTheta'} <= set(h)
  assert 'initPoint' not in h
  fans = store.hits()
  assert fans.supportsFanMath() and fans.fanCount() == 2 and fans.raysPerFan() == 20
  assert fans.fanNeighborDists().shape[0] == 3 and np.isfinite(fans.fanCenter()).all()
  # initial conditions of device-generated rays, recomputed from the counter-based stream
  for k in ('StoreHitInitPoint', 'StoreHitInitDirection', 'StoreHitInitPower', 'StoreHitInitWavelength'):
    setattr(st, k, True)
  store = runSimulation(doc, 'singletrue', seed=77)
  h = store.hits().hits
  m = len(h['points'])
  assert m > 90 and h['initPoint'].shape == (m, 3) and h['initDirection'].shape == (m, 3)
  assert np.allclose(h['initPoint'], 0) and np.all(h['initPower'] == 1) and np.all(h['initWavelength'] == 500)
  th, ph = h['initTheta'], h['initPhi']
  want = np.stack([np.sin(th) * np.sin(ph), -np.sin(th) * np.cos(ph), np.cos(th)], axis=1)   # source at the origin, +z
  assert np.abs(h['initDirection'] - want).max() < 1e-12
  assert 'fanIndex' not in h                     # Monte-Carlo rays carry no fan metadata
  # pseudo-random modes: histogram-thinned host draws traced on the device
  store = runSimulation(doc, 'singlepseudo')
  assert store.totalTracedRays == 100 and store.totalIterations == 1 and len(store.hits()) > 90
  st.EndAfterRays = '1e3'
  store = runSimulation(doc, 'pseudo', pseudoIterationsPerLaunch=4)
  assert 1000 < store.totalTracedRays <= 1400 and len(store.hits()) > 900


def test_freecad_document_property_api(tmp_path):
  """property round trips of test/20-freecad-document/2-from-fcstd-folder.py, without FreeCAD"""
  import shutil
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument
  src = os.path.join(SCENES, 'GettingStarted.FCStd')
  with FreecadDocument(src, workInTempCopy=True) as f:
    assert f.Sphere.Radius.get() == pytest.approx(9.8275862)
    f.Sphere.Radius = 10.5
    assert float(f.Sphere.Radius) == 10.5
    f.OpticalSimulationSettings.EndAfterRays = '1e3'
    assert f.OpticalSimulationSettings.EndAfterRays.get() == '1e3'
    label = f.document().OpticalPointSource._props['Label']
    assert getattr(f, label).PowerDensity.get() == 'exp(-theta^2/0.01)'           # resolved by label
    assert f.OpticalPointSource.PowerDensity == getattr(f, label).PowerDensity     # and by internal name
    # attribute paths below a Placement, as in a FreeCAD shell (angle in radians)
    f.Sphere.Placement.Base.z = -7.5
    assert np.allclose(f.Sphere.Placement.Base.get(), [0, 0, -7.5]) and float(f.Sphere.Placement.Base.z) == -7.5
    f.Cube.Placement.Rotation.Angle = 0.25
    axis, angle = f.Cube.Placement.get().axisAngle()
    assert abs(angle - 0.25) < 1e-12 and abs(float(f.Cube.Placement.Rotation.Angle) - 0.25) < 1e-12
    f.Cube.Placement.Rotation.Axis = (0, 0, 1)
    assert np.allclose(f.Cube.Placement.get().Rotation, [[np.cos(.25), -np.sin(.25), 0], [np.sin(.25), np.cos(.25), 0], [0, 0, 1]])
    with pytest.raises(AttributeError):
      f.Cube.Placement.Rotation.Nonsense = 1
    with pytest.raises(AttributeError):
      f.NoSuchObject
    tmp = f._tmp
  assert not os.path.exists(tmp)


@pytest.mark.gpu
def test_freecad_document_run_simulation(native_lib):
  """examples/1-getting-started/optimize-spotsize.ipynb cell 9 pattern"""
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument
  with FreecadDocument(os.path.join(SCENES, 'GettingStarted.FCStd'), workInTempCopy=True) as f:
    f.OpticalSimulationSettings.EndAfterRays = '2e4'
    sizes = []
    for r in (9.5, 10.25, 11.0):
      f.Sphere.Radius = r
      raw = f.runSimulation('true')
      h = raw.loadHits('*')
      assert len(h) > 1.9e4
      p = h.points()
      sizes.append(float(np.sqrt(((p - np.median(p, axis=0))**2).sum(1).mean())))
    assert len(f.rawFolders()) == 3
    assert sizes[1] < sizes[0] and sizes[1] < sizes[2]     # best focus near R = 10.27
    hist = f.latestRawFolder().loadHits('*').histogram(bins=30)
    assert hist.hist.sum() == len(h)


def test_global_info_progress_and_status_files(tmp_path):
  """the rest of the run-folder contract (results_store.py:333-336, 508-538;
  simulation_loop.py:174-269; freecad_elements/__init__.py:48-115)"""
  import pickle
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.scene import bake, open_fcstd
  from freecad.optics_design_workbench_amd.simulation.results_store import RawFolder, SimulationResults
  doc = open_fcstd(os.path.join(SCENES, 'global-placement-main.FCStd'))
  info = bake.collectGlobalInfo(doc)
  assert set(info) == {'activeSimulationSettings', 'lightSources', 'opticalObjects'}
  assert info['activeSimulationSettings']['Name'] == 'OpticalSimulationSettings'
  (src,) = info['lightSources']
  assert src['name'] == 'OpticalPointSource' and 'PowerDensity' in src['properties']
  assert 'Placement' not in src['properties'] and 'Proxy' not in src['properties']
  (rep,) = src['placementPathsAndMatrices']
  assert rep['path'][-1] == 'OpticalPointSource' and np.allclose(rep['gpM'][:3, 3], [5, 0, -11])
  assert np.allclose(rep['gpM'] @ rep['gpMi'], np.eye(4)) and np.allclose(rep['pM'] @ rep['pMi'], np.eye(4))
  assert np.allclose(rep['pM'][:3, 3], [0, 0, 4])                      # the source's own placement
  assert [o['properties']['OpticalType'] for o in info['opticalObjects']] == ['Mirror', 'Lens', 'Mirror', 'Absorber']
  pickle.dumps(info)                                                   # everything exported is picklable

  store = SimulationResults('true', resultsPath=str(tmp_path / 'x.OpticsDesign'), endAfterRays=1e3)
  store.dumpGlobalInfo(info)
  store.dumpGlobalInfo(dict(other=1))                                  # written once, never overwritten
  for i in range(14):
    store.incrementRayCount(100)
    store.incrementIterationCount()
    store.dumpProgress()
  raw = RawFolder(store.runFolderPath())
  assert raw.loadGlobalInfo()['lightSources'][0]['name'] == 'OpticalPointSource'
  prog = raw.loadProgress()
  assert prog['totalTracedRays'] == 1400 and prog['totalIterations'] == 14 and prog['endAfterRays'] == 1e3
  assert prog['simulationType'] == 'true' and prog['endAfterHits'] == np.inf
  files = sorted(os.listdir(os.path.join(store.runFolderPath(), 'progress')))
  assert files == [f'master-{i:09d}' for i in range(4, 14)]            # older summaries are pruned
  assert len(raw.loadHits('*')) == 0                                   # progress files are not hit files
  base = str(tmp_path / 'x.OpticsDesign')
  store.setStatus('simulation-is-running', True)
  store.setStatus('simulation-is-running', True)
  assert os.path.exists(os.path.join(base, 'simulation-is-running'))
  store.setStatus('simulation-is-running', False)
  assert not os.path.exists(os.path.join(base, 'simulation-is-running'))


def test_raw_folder_helpers(tmp_path, oracle):
  """rawFolders / rawFolderByIndex / RawFolderRange / RawFolder.tree
  (jupyter_utils/freecad_document.py:1341-1539) on runs of the oracle-backed loop"""
  import shutil
  from conftest import SCENES
  from oracle_tracer import OracleTracer
  from freecad.optics_design_workbench_amd import jupyter_utils
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from freecad.optics_design_workbench_amd.simulation import resultsFolderPath, runSimulation
  path = str(tmp_path / 'GettingStarted.FCStd')
  shutil.copy(os.path.join(SCENES, 'GettingStarted.FCStd'), path)
  doc = open_fcstd(path)
  doc.OpticalPointSource.RecordRays = True
  stores = [runSimulation(doc, 'singletrue', resultsPath=resultsFolderPath(path), tracer=OracleTracer(), seed=s)
            for s in (1, 2, 3)]
  for base in (str(tmp_path), resultsFolderPath(path), os.path.join(resultsFolderPath(path), 'raw', 'simulation-run-000001')):
    rng = jupyter_utils.rawFolders(base)
    assert isinstance(rng, jupyter_utils.RawFolderRange) and len(rng) == 3
  assert [os.path.basename(p) for p in rng.paths()] == [f'simulation-run-{i:06d}' for i in range(3)]
  assert len(rng.loadHits('*')) == sum(len(s.hits()) for s in stores)
  assert len(rng[1:].loadHits('*')) == sum(len(s.hits()) for s in stores[1:])
  assert len(rng.loadRays()) == 300 and isinstance(rng[0], jupyter_utils.RawFolder)
  assert jupyter_utils.rawFolderByIndex(1, base).path() == rng[1].path()
  assert jupyter_utils.rawFolderByIndex(-1, base).path() == jupyter_utils.latestRawFolder(base).path() == rng[2].path()
  with pytest.raises(ValueError):
    jupyter_utils.rawFolderByIndex(7, base)
  tree = rng[0].tree()
  src = tree['source-OpticalPointSource']
  assert src['<1 ray files>'] is None and any('<1 hit files>' in v for v in src.values() if isinstance(v, dict))
  rng[0].printTree()
  assert len(jupyter_utils.rawFolders(str(tmp_path / 'nowhere'))) == 0 if os.path.isdir(tmp_path / 'nowhere') else True
  # transforms of global-info matrices
  m = np.eye(4); m[:3, 3] = (1, 2, 3); m[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
  p = np.array([[1.0, 0, 0], [0, 2.0, 0]])
  assert np.allclose(jupyter_utils.applyTransformation(p, m), [[1, 3, 3], [-1, 2, 3]])
  assert np.allclose(jupyter_utils.applyTransformationWithoutTranslation(p, m), [[0, 1, 0], [-2, 0, 0]])
  with jupyter_utils.FreecadDocument(path, workInTempCopy=True) as f:
    f.disableFastMode()
    assert f.isWorkInTempCopy() and f.path().endswith('GettingStarted.FCStd') and f.resultsPath().endswith('.OpticsDesign')
    assert f.Sphere.Radius.getFloat() == float(f.Sphere.Radius) and f.OpticalSimulationSettings.RaysPerIteration.getInt() == 100


def test_histogram_plots():
  """Histogram.plot / plotByAzimuth (jupyter_utils/histogram.py:91-166) on an off-screen canvas"""
  import matplotlib
  matplotlib.use('Agg')
  import matplotlib.pyplot as plt
  from freecad.optics_design_workbench_amd.jupyter_utils import Histogram
  rng = np.random.default_rng(3)
  x, y = rng.normal(0, 1, 20000), rng.normal(0, 2, 20000)
  h = Histogram(x, y, planeNormal=(0, 0, 1), xInPlaneVec=(1, 0, 0), bins=40)
  assert h.scaledHist().max() == 1.0 and np.allclose(h.scaledHist(None), h.hist.T)
  plt.figure()
  mesh = h.plot()
  assert mesh.get_array().max() == 1.0 and plt.gca().get_xlabel() == 'projected $x$'
  hp = Histogram(x, y, planeNormal=(0, 0, 1), xInPlaneVec=(1, 0, 0), binCoords='polar', radius=4, bins=(12, 20))
  plt.figure()
  hp.plot(cbar=None, scale=None)
  assert plt.gca().name == 'polar'
  plt.figure()
  hp.plotByAzimuth()
  assert len(plt.gca().get_lines()) == 12
  from freecad.optics_design_workbench_amd.jupyter_utils import Hits
  pts = np.stack([x[:500], y[:500], np.full(500, 7.0)], axis=1)
  hits = Hits(dict(points=pts, directions=np.tile([0, 0, 1.0], (500, 1)), fanIndex=np.arange(500) % 3))
  plt.figure()
  sc = hits.plot(hueKey='fanIndex')
  assert len(sc.get_offsets()) == 500 and 'plane normal' in plt.gca().get_title()
  assert Hits({}).plot() is None
  plt.close('all')


@pytest.mark.gpu
def test_recorded_rays_and_surface_fans_through_the_document_api(native_lib):
  """the whole stack on the device: RecordRays -> `*-rays.pkl` -> loadRays; fan mode of a surface
  source (test/21-simulation-modes/main.FCStd) and of an imported shape's faces (test/80)"""
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument
  with FreecadDocument(os.path.join(SCENES, 'GettingStarted.FCStd'), workInTempCopy=True) as f:
    f.OpticalPointSource.RecordRays = True
    f.OpticalSimulationSettings.EndAfterRays = '250'
    raw = f.runSimulation('true')
    rays = raw.loadRays()
    assert len(rays) == raw.loadProgress()['totalTracedRays'] > 250
    ends = {tuple(r['points'][-1]) for r in rays}
    assert all(tuple(p) in ends for p in raw.loadHits('*').points())
    assert all(len(r['media']) == len(r['powers']) == len(r['points']) - 1 for r in rays)
    assert any('OpticalLensGroup' in r['media'] for r in rays)
  with FreecadDocument(os.path.join(SCENES, 'simulation-modes-main.FCStd'), workInTempCopy=True) as f:
    raw = f.runSimulation('fans')
    assert raw.loadProgress()['totalTracedRays'] == 121 and len(raw.loadHits('*')) > 20
  with FreecadDocument(os.path.join(SCENES, 'imported-stepfile-as-surface-source.FCStd'), workInTempCopy=True) as f:
    f.OpticalSimulationSettings.EndAfterRays = '2e4'
    assert 80 < f.runSimulation('fans').loadProgress()['totalTracedRays'] < 160
    assert f.runSimulation('true').loadProgress()['totalTracedRays'] > 2e4


def test_flushes_written_in_the_background(tmp_path):
  """flush(wait=False) -- the run loop's -- hands the files to the writer thread; drain() waits for them, and
  whatever the writer met is raised there"""
  from freecad.optics_design_workbench_amd.simulation import results_store as rs
  res = rs.resultsFolderPath(str(tmp_path / 'proj.FCStd'))
  store = rs.SimulationResults('true', resultsPath=res)
  for k in range(5):
    _fill(store, 2000, k)
    store.flush(wait=False)
  store.drain()
  folder = os.path.join(store.runFolderPath(), 'source-src', 'object-OpticalAbsorberGroup')
  files = sorted(os.listdir(folder))
  assert len(files) == 5
  total = 0
  for f in files:
    d = pickle.load(open(os.path.join(folder, f), 'rb'))          # plain pickle.load, as the reference reads them
    assert d['points'].shape == (2000, 3)
    total += len(d['powers'])
  assert total == 10000 and len(rs.latestRawFolder(res).loadHits('*')) == 10000
  # an error of the writer surfaces at the next drain
  _fill(store, 10, 9)
  import shutil
  shutil.rmtree(store.runFolderPath())
  real_makedirs = os.makedirs
  try:
    os.makedirs = lambda *a, **k: None                          # the folder is gone: open() fails in the writer
    store.flush(wait=False)
  finally:
    os.makedirs = real_makedirs
  with pytest.raises(OSError):
    store.drain()


def test_writer_thread_ends_with_the_run(tmp_path):
  import threading
  from freecad.optics_design_workbench_amd.simulation import results_store as rs
  res = rs.resultsFolderPath(str(tmp_path / 'proj.FCStd'))
  before = threading.active_count()
  store = rs.SimulationResults('true', resultsPath=res)
  _fill(store, 100, 0)
  store.flush(wait=False)
  store.drain(stop=True)
  assert threading.active_count() == before
  _fill(store, 100, 1)                 # a later flush starts a new writer
  store.flush()
  store.drain(stop=True)
  assert threading.active_count() == before
  assert len(rs.latestRawFolder(res).loadHits('*')) == 200


def test_a_failed_write_ends_the_run_at_the_next_flush(tmp_path, monkeypatch):
  """the writer thread meets an error (disk full, permissions): the flush after it raises -- the run loop's
  `flush(wait=False)` included --, not only the drain at the end of the run; no half-written file is left
  under a name readers look for"""
  from freecad.optics_design_workbench_amd.simulation import results_store as rs
  res = rs.resultsFolderPath(str(tmp_path / 'proj.FCStd'))
  store = rs.SimulationResults('true', resultsPath=res)
  real_dump = pickle.dump
  state = dict(fail=True)

  def dump(obj, f, **kw):
    if state['fail'] and getattr(f, 'name', '').endswith('-hits.pkl.tmp'):
      f.write(b'half a pickle')
      raise OSError(28, 'No space left on device')
    return real_dump(obj, f, **kw)
  monkeypatch.setattr(rs.pickle, 'dump', dump)
  _fill(store, 10, 0)
  store.flush(wait=False)                      # handed to the writer, which fails
  store._writeQueue.join()
  _fill(store, 10, 1)
  with pytest.raises(OSError):
    store.flush(wait=False)                    # the run loop learns of it here
  folder = os.path.join(store.runFolderPath(), 'source-src', 'object-OpticalAbsorberGroup')
  assert not [f for f in os.listdir(folder) if f.endswith('-hits.pkl')]
  state['fail'] = False
  _fill(store, 10, 2)
  store.flush()
  store.drain(stop=True)
  assert len([f for f in os.listdir(folder) if f.endswith('-hits.pkl')]) == 1
