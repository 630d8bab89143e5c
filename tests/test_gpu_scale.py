"""Full-size and statistical checks (BASELINE sizes; size-independent
properties where the oracle cannot follow)."""
import numpy as np
import pytest

from conftest import project

pytestmark = pytest.mark.gpu
SEED = 0x0D15EA5E


@pytest.fixture(scope='module')
def tracer(native_lib):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  tr = Tracer(0)
  yield tr
  tr.close()


def kl(p, q):
  p = p.astype(float).ravel() / p.sum()
  q = q.astype(float).ravel() / q.sum()
  m = (p > 0) & (q > 0)
  return float(np.sum(p[m] * np.log(p[m] / q[m])))


def setup(tracer, name, nx=64, capacity=0):
  from freecad.optics_design_workbench_amd import scenes
  pr = project(name)
  det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=nx, ny=nx, window=2.0,
                             toward=pr.source.xform[[3, 7, 11]])
  if name == 'lensesAndMirrors':
    det['origin'] = [-68.8618, 0.0, 73.0]          # beam centre on the absorber
  tracer.setScene(pr.scene)
  tracer.setSource(pr.source)
  tracer.setLimits(pr.limits)
  tracer.setDetector(det)
  tracer.reserveHits(capacity)
  tracer.reset()
  return pr, det


def test_histogram_kl_gpu_vs_cpu_reference(tracer, oracle):
  """north star: detector histograms match the CPU reference to < 1 % KL.
  Same seed: identical.  Different seeds (independent samples): KL stays
  at the sampling-noise level (bins/2N), far below 1 %."""
  pr, det = setup(tracer, 'lensesAndMirrors', nx=32)
  n = 2_000_000
  tracer.trace(0, n, SEED, record_hits=False)
  tracer.sync()
  h_gpu = tracer.histogram()
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 0, n, SEED, det=det, flags=2, nthreads=0)
  assert np.array_equal(h_gpu, ref['hist'])
  ref2 = oracle.trace(pr.scene, pr.source, pr.limits, 0, n, SEED + 1, det=det, flags=2, nthreads=0)
  assert kl(h_gpu, ref2['hist']) < 0.01
  assert kl(h_gpu, ref2['hist']) < 5 * (32 * 32) / (2 * n)


def test_c3_full_size_properties(tracer, oracle):
  """1e8 rays (BASELINE configs[2]): conservation laws + the first rows equal
  the oracle's + linearity of the histogram in the ray range"""
  n = 100_000_000
  pr, det = setup(tracer, 'lensesAndMirrors', nx=64, capacity=n + 1024)
  tracer.trace(0, n, SEED)
  tracer.sync()
  c = tracer.counters()
  assert c['traced_rays'] == n
  assert c['escaped'] + c['died'] + c['capped'] == n
  assert c['hits_dropped'] == 0 and tracer.hitCount() == c['recorded_hits']
  h_all = tracer.histogram()
  assert int(h_all.sum()) + c['hist_overflow'] == c['recorded_hits']
  assert 6.9 * n < c['segments'] < 7.1 * n
  # additivity: hist[0,n) == hist[0,n/2) + hist[n/2,n)
  tracer.reset()
  tracer.trace(0, n // 2, SEED, record_hits=False)
  tracer.sync()
  h1 = tracer.histogram()
  tracer.reset()
  tracer.trace(n // 2, n - n // 2, SEED, record_hits=False)
  tracer.sync()
  assert np.array_equal(h_all, h1 + tracer.histogram())
  # a window in the middle of the job against the oracle
  tracer.reserveHits(300000)
  tracer.reset()
  tracer.trace(50_000_000, 100000, SEED)
  tracer.sync()
  g = tracer.hits()
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 50_000_000, 100000, SEED, nthreads=0)
  assert np.array_equal(g['tag'], ref['hits']['tag'])
  assert np.abs(g['point'] - ref['hits']['point']).max() < 1e-9


def test_c2_minimal_ten_million(tracer, oracle):
  """BASELINE configs[1]: Gaussian source + single detector, 1e7 rays"""
  n = 10_000_000
  pr, det = setup(tracer, 'minimal', nx=64, capacity=n + 1024)
  tracer.trace(0, n, SEED)
  tracer.sync()
  c = tracer.counters()
  assert c['traced_rays'] == n and c['recorded_hits'] == n and c['segments'] == n
  h = tracer.hits()
  r = np.hypot(h['point'][:, 0], h['point'][:, 1])
  # exp(-theta^2/sigma^2): <r^2> = (15 mm * sigma)^2 for sigma = 0.01
  assert np.mean(r**2) == pytest.approx((15 * 0.01)**2, rel=2e-3)
  assert np.abs(h['point'][:, 2] - 15.0).max() < 1e-12


def test_more_rays_than_32_bits_in_one_launch(tracer):
  """maximum sizes: one launch of 2^32 + 54321 rays (histogram only: the rows of such a launch would be 275 GB) --
  every ray traced and binned, and the same histogram and counters as the same index range in two launches split
  at 2^32 (ray indices, chunk arithmetic and counters are 64-bit throughout; a ray depends on its index only)"""
  n = (1 << 32) + 54321
  setup(tracer, 'minimal', nx=128)
  tracer.trace(7, n, SEED, record_hits=False)
  tracer.sync()
  c = tracer.counters()
  assert c['traced_rays'] == n and c['recorded_hits'] == n and c['segments'] == n and c['hits_dropped'] == 0
  one = tracer.histogram().copy()
  assert int(one.sum()) + c['hist_overflow'] == n
  tracer.reset()
  tracer.trace(7, 1 << 32, SEED, record_hits=False)
  tracer.trace(7 + (1 << 32), 54321, SEED, record_hits=False)
  tracer.sync()
  assert tracer.counters() == c
  assert np.array_equal(tracer.histogram(), one)


def test_c4_huge_array_statistics(tracer, oracle):
  """BASELINE configs[3] scene on one GPU: chaotic, so statistics only:
  hit fraction and mean segment count agree with the oracle within 3 sigma"""
  pr = project('hugeArray')
  tracer.setScene(pr.scene)
  tracer.setSource(pr.source)
  tracer.setLimits(pr.limits)
  tracer.setDetector(None)
  n = 2_000_000
  tracer.reserveHits(n)
  tracer.reset()
  tracer.trace(0, n, SEED)
  tracer.sync()
  c = tracer.counters()
  m = 200_000
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 10_000_000, m, SEED, nthreads=0)['counters']
  p_gpu, p_ref = c['recorded_hits'] / n, ref['recorded_hits'] / m
  sigma = np.sqrt(p_ref * (1 - p_ref) * (1 / n + 1 / m))
  assert abs(p_gpu - p_ref) < 4 * sigma
  assert c['segments'] / n == pytest.approx(ref['segments'] / m, rel=0.02)
  assert c['escaped'] + c['died'] + c['capped'] == n


def test_c4_full_per_gpu_share(tracer):
  """BASELINE configs[3]: the 1.25e8-ray share one GPU traces of the 1e9-ray hugeArray job, in the
  bench's launch shape (hit rows + 1024^2 histogram in projection): conservation laws, and the
  histogram of the whole range equals the sum over two half ranges (what the RCCL reduce of two
  ranks adds up)"""
  pr = project('hugeArray')
  gi = pr.scene.group_index('OpticalAbsorberGroup')
  det = dict(group=gi, origin=[-0.5, -0.5, 61.0], ex=[1.0, 0.0, 0.0], ey=[0.0, 1.0, 0.0],
             x_lo=-25.0, x_hi=25.0, y_lo=-25.0, y_hi=25.0, nx=1024, ny=1024)
  n = 125_000_000
  tracer.setScene(pr.scene)
  tracer.setSource(pr.source)
  tracer.setLimits(pr.limits)
  tracer.setDetector(det)
  tracer.reserveHits(n // 2)
  tracer.reset()
  tracer.trace(0, n, SEED)
  tracer.sync()
  c = tracer.counters()
  assert c['traced_rays'] == n and c['escaped'] + c['died'] + c['capped'] == n
  assert c['hits_dropped'] == 0 and tracer.hitCount() == c['recorded_hits']
  assert 0.13 * n < c['recorded_hits'] < 0.19 * n and 2.6 * n < c['segments'] < 3.1 * n
  h_all = tracer.histogram()
  assert int(h_all.sum()) + c['hist_overflow'] == c['recorded_hits']
  assert c['hist_overflow'] < 1e-3 * c['recorded_hits']          # the window covers the array's footprint
  parts = []
  for first, m in ((0, n // 2), (n // 2, n - n // 2)):
    tracer.reset()
    tracer.trace(first, m, SEED, record_hits=False)
    tracer.sync()
    parts.append((tracer.histogram(), tracer.counters()))
  assert np.array_equal(h_all, parts[0][0] + parts[1][0])
  for k in ('segments', 'recorded_hits', 'escaped', 'died', 'capped'):
    assert c[k] == parts[0][1][k] + parts[1][1][k], k


def test_c5_full_radius_sweep(tracer):
  """BASELINE configs[4]: 64 radii x 1e7 rays through `sweep.parameterSweep` with the notebook's
  calcFwhm on the hit rows in HBM.  Size-independent checks: every radius traced exactly 1e7
  rays, every spot size is a positive number; the rms spot radius (second moment of the same
  rows) is a smooth curve with a single minimum inside the range (the lens focuses onto the
  detector near R = 10.3 mm)"""
  import os
  from conftest import SCENES
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from freecad.optics_design_workbench_amd.simulation import sweep
  doc = open_fcstd(os.path.join(SCENES, 'GettingStarted.FCStd'))
  radii = np.linspace(9, 11, 64)

  def setRadius(d, r):
    d.Sphere.Radius = r
  res = sweep.parameterSweep(doc, setRadius, radii, rays=10_000_000, seed=SEED, tracer=tracer,
                             measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot))
  assert res.tracedRays == 64 * 10_000_000
  fwhm, rms = res.columns['fwhm'], res.columns['rms']
  # the notebook's FWHM: a positive number, or nan where its fit finds no half-maximum (near the
  # focus, with this many hits; see sweep.rmsSpot) -- far from the focus it is defined
  assert np.all((fwhm > 0) | np.isnan(fwhm)) and np.isfinite(fwhm[:5]).all() and np.isfinite(fwhm[-5:]).all()
  k = int(np.argmin(rms))
  assert 10.0 < radii[k] < 10.6, radii[k]
  assert np.all(np.diff(rms[:k + 1]) < 0) and np.all(np.diff(rms[k:]) > 0)      # one valley
  assert rms[k] < 0.25 * max(rms[0], rms[-1])
