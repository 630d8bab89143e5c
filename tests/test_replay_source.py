"""ReplaySourceProxy (freecad_elements/replay_source.py): hits of an earlier run
become the rays of the next one."""
import os

import numpy as np
import pytest

from freecad.optics_design_workbench_amd.freecad_elements import make, replay_source
from freecad.optics_design_workbench_amd.scene import Document, Placement, bake
from freecad.optics_design_workbench_amd.simulation.results_store import SimulationResults


def _recorded_run(tmp_path, n=500, seed=1):
  """a run folder with two hit files (the second one with wavelengths)"""
  rs = np.random.RandomState(seed)
  store = SimulationResults('true', resultsPath=str(tmp_path / 'first.OpticsDesign'))
  rows = []
  for k, extra in enumerate(({}, dict(wavelength=np.full(n, 633.0)))):
    p = rs.normal(0, 1, (n, 3)) + [0, 0, 10]
    d = rs.normal(0, 0.05, (n, 3)) + [0, 0, 1]
    d /= np.linalg.norm(d, axis=1)[:, None]
    pw = rs.uniform(0.2, 1, n)
    store.addRayHits('Src', 'Src', f'Obj{k}', f'Obj{k}', p, d, pw, np.ones(n), **extra)
    rows.append((p, d, extra.get('wavelength', np.ones(n)), pw))
  store.flush()
  return store.runFolderPath(), rows


def _key(p, d, wl, pw):
  a = np.concatenate([p, d, wl[:, None], pw[:, None]], axis=1)
  return a[np.lexsort(a.T[::-1])]


def test_stock_is_every_recorded_row_once_with_placement(tmp_path):
  folder, rows = _recorded_run(tmp_path)
  doc = Document()
  pl = Placement(base=(1, 2, 3), quat=(0, np.sin(0.3), 0, np.cos(0.3)))
  src = make.makeReplaySource(doc, folder, placement=pl)
  b = replay_source.bakeReplaySource(doc, src, seed=5)
  assert len(b.origins) == 1000 and b.remaining == 1000
  p = np.concatenate([r[0] for r in rows]); d = np.concatenate([r[1] for r in rows])
  wl = np.concatenate([r[2] for r in rows]); pw = np.concatenate([r[3] for r in rows])
  R, t = pl.m[:3, :3], pl.m[:3, 3]
  want = _key(p @ R.T + t, d @ R.T, wl, pw)
  got = _key(b.origins, b.directions, b.wavelengths, b.powers)
  assert np.allclose(got, want, atol=1e-12)
  # shuffled, reproducibly
  b2 = replay_source.bakeReplaySource(doc, src, seed=5)
  assert np.array_equal(b.origins, b2.origins) and not np.array_equal(b.origins[:50], (p @ R.T + t)[:50])
  # iterations consume the stock; rewinding restores it (onInitializeSimulation)
  o, *_ = b.take(300)
  assert len(o) == 300 and b.remaining == 700
  o, *_ = b.take(900)
  assert len(o) == 700 and b.remaining == 0 and len(b.take(10)[0]) == 0
  b.rewind()
  assert b.remaining == 1000


def test_errors(tmp_path):
  doc = Document()
  with pytest.raises(RuntimeError, match='replay directory'):
    replay_source.bakeReplaySource(doc, make.makeReplaySource(doc, ''))
  with pytest.raises(RuntimeError, match='exist'):
    replay_source.bakeReplaySource(doc, make.makeReplaySource(doc, str(tmp_path / 'nope'), name='R2'))
  os.makedirs(tmp_path / 'empty')
  with pytest.raises(RuntimeError, match='any ray hit datafile'):
    replay_source.bakeReplaySource(doc, make.makeReplaySource(doc, str(tmp_path / 'empty'), name='R3'))


def test_reference_document_loads():
  """test/50-old-tests/replay.FCStd's source object is recognised (its
  ReplayFromDir points into the author's home directory)"""
  ref = '/root/reference/test/50-old-tests/replay.FCStd'
  if not os.path.exists(ref):
    pytest.skip('reference not mounted')
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  doc = open_fcstd(ref)
  (src,) = bake.lightSources(doc)
  assert src.ProxyClass == 'ReplaySourceProxy' and src.ReplayFromDir.endswith('simulation-run-000000')
  with pytest.raises(RuntimeError, match='exist'):
    replay_source.bakeReplaySource(doc, src)


@pytest.mark.gpu
def test_replay_run_on_device(native_lib, oracle, tmp_path):
  from freecad.optics_design_workbench_amd.simulation import runSimulation
  folder, rows = _recorded_run(tmp_path, n=2000)
  doc = Document()
  make.makeMirror(doc, [make.makeBox(doc, 'M', 200, 200, 1, base=(-100, -100, 60), quat=(0, np.sin(0.2), 0, np.cos(0.2)))])
  make.makeGrating(doc, [make.makeBox(doc, 'G', 400, 400, 1, base=(-200, -200, -80))], RecordHits=True,
                   GratingLinesPerMillimeter=300.0, GratingLinesOrientation=np.array([1.0, 0, 0]))
  make.makeAbsorber(doc, [make.makeBox(doc, 'A', 2000, 2000, 1, base=(-1000, -1000, 400))])
  make.makeSimulationSettings(doc, EndAfterRays='inf', EndAfterIterations='inf', RaysPerIteration=100.0,
                              MaxRayLength=1e4)
  src = make.makeReplaySource(doc, folder)
  with pytest.warns(UserWarning, match='ran out of rays'):
    store = runSimulation(doc, 'true', seed=9, endIf=lambda s: False, raysPerLaunch=1500)
  assert store.totalTracedRays == 4000              # the stock, once
  h = store.hits().hits
  # same stock through the oracle, one wavelength at a time
  b = replay_source.bakeReplaySource(doc, src, seed=9)
  sc, lim = bake.bakeScene(doc, src), bake.bakeLimits(doc, src)
  n_ref, pts = 0, []
  for w in np.unique(b.wavelengths):
    sel = b.wavelengths == w
    r = oracle.trace_rays(sc, lim, b.origins[sel], b.directions[sel], b.powers[sel], wavelength=w)
    n_ref += len(r['hits']); pts.append(r['hits']['point'])
  assert len(h['points']) == n_ref > 3000
  def ordered(p):          # order by coordinates rounded well above the device/oracle rounding noise
    return p[np.lexsort(np.round(p, 4).T)]
  got, want = ordered(h['points']), ordered(np.concatenate(pts))
  assert (np.abs(got - want).max(axis=1) < 1e-7).mean() > 0.999
  # fan mode places no rays for a replay source (replay_source.py:131-134)
  assert runSimulation(doc, 'fans').totalTracedRays == 0
