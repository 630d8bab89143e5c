"""GPU vs oracle on hand-built scenes that exercise every primitive kind,
boolean trimming, flipped normals, gratings and limits -- with explicit rays
(odw_trace_rays), all groups recording, so whole trajectories are compared."""
import copy

import numpy as np
import pytest

from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import Document, Placement, bake

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope='module')
def tracer(native_lib):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  tr = Tracer(0)
  yield tr
  tr.close()


def quat(axis, deg):
  a = np.asarray(axis, float) / np.linalg.norm(axis)
  h = np.radians(deg) / 2
  return (*(a * np.sin(h)), np.cos(h))


def build(groups, settings=None):
  doc = Document()
  for kind, elems, props in groups:
    make.makeOpticalGroup(doc, kind, elems(doc), **props)
  make.makeSimulationSettings(doc, **(settings or {}))
  src = make.makePointSource(doc)
  sc = bake.bakeScene(doc, src)
  sc.group_record = np.ones_like(sc.group_record)
  return sc, bake.bakeLimits(doc, src)


def aimed_rays(n, targets, spread, seed, radius=60.0):
  rs = np.random.RandomState(seed)
  o = rs.normal(0, 1, (n, 3))
  o = o / np.linalg.norm(o, axis=1)[:, None] * radius
  t = np.asarray(targets, float)[rs.randint(0, len(targets), n)] + rs.normal(0, spread, (n, 3))
  d = t - o
  return o, d / np.linalg.norm(d, axis=1)[:, None]


def run_both(tracer, oracle, sc, lim, o, d):
  tracer.setScene(sc)
  tracer.setLimits(lim)
  tracer.setDetector(None)
  tracer.reserveHits(len(o) * (lim.max_intersections + 1))
  tracer.reset()
  tracer.traceRays(o, d)
  tracer.sync()
  g, gc = tracer.hits(), tracer.counters()
  ref = oracle.trace_rays(sc, lim, o, d)
  return g, gc, ref['hits'], ref['counters']


def ordinal(tags):
  """index of each row within its ray (rows are sorted by ray, then bounce)"""
  r = (tags & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
  _, first, inv = np.unique(r, return_index=True, return_inverse=True)
  return np.arange(len(r)) - first[inv]


def path_tolerance(tags, tol=TOL, growth=50.0, cap=1e-4):
  """curved mirrors and lenses amplify a rounding difference at every
  interaction (a sphere of radius R hit at distance L: ~2L/R); the k-th
  intersection of a ray is compared at tol * growth**(k-1), first two at tol"""
  k = np.maximum(ordinal(tags) - 1, 0)
  return np.minimum(tol * growth**k, cap)[:, None]


def assert_same(g, gc, r, rc, tol=TOL):
  assert gc == rc
  assert np.array_equal(g['tag'], r['tag'])           # ray, group, isEntering of every intersection
  assert_close_rows(g, r, np.ones(len(r), dtype=bool), np.ones(len(r), dtype=bool), tol)
  assert np.abs(g['power'] - r['power']).max() < 1e-9


def assert_close_rows(g, r, sg, sr, tol):
  """coordinates: >= 99.9 % of the intersections within the path tolerance,
  all of them within 1e-4 mm.  (The rest are grazing hits on curved faces:
  t is ill-conditioned there, error ~ rounding / cos(incidence), on the
  device and in the oracle alike.)"""
  t = path_tolerance(r['tag'], tol)[sr]
  dp = np.abs(g['point'][sg] - r['point'][sr])
  dd = np.abs(g['direction'][sg] - r['direction'][sr])
  assert np.all(dp < t, axis=1).mean() > 0.999 and np.all(dd < t, axis=1).mean() > 0.999
  assert dp.max() < 1e-4 and dd.max() < 1e-4


def assert_same_short_paths(g, gc, r, rc, n_rays, max_len=12, tol=TOL):
  """like assert_same, for scenes where a few rays get trapped (total internal
  reflection inside a torus lens: ~100 bounces): those trajectories amplify
  rounding differences until they differ.  Every ray whose path has at most
  `max_len` recorded intersections on both sides must agree exactly; at most
  0.1 % of the rays may fall outside that class or differ in length."""
  m48 = np.uint64(0xFFFFFFFFFFFF)
  gr, rr = (g['tag'] & m48).astype(np.int64), (r['tag'] & m48).astype(np.int64)
  cg, cr = np.bincount(gr, minlength=n_rays), np.bincount(rr, minlength=n_rays)
  ok = (cg == cr) & (cg <= max_len)
  print('rays with different path length:', int((cg != cr).sum()), ' long paths:', int((cr > max_len).sum()),
        ' of', n_rays)
  assert (cg != cr).mean() < 2e-3
  assert ok.mean() > 0.99
  sg, sr = ok[gr], ok[rr]
  assert np.array_equal(g['tag'][sg], r['tag'][sr])
  assert_close_rows(g, r, sg, sr, tol)
  assert gc['traced_rays'] == rc['traced_rays']


def test_torus(tracer, oracle):
  """absorber + lens tori: the device marches the distance function, the
  oracle isolates the quartic's roots"""
  sc, lim = build([
      ('Absorber', lambda d: [make.makeTorus(d, 'T1', 10, 2, base=(0, 0, 0))], {}),
      ('Lens', lambda d: [make.makeTorus(d, 'T2', 6, 1.5, base=(0, 25, 3), quat=quat((1, 0, 0), 40))],
       dict(RefractiveIndex=1.5)),
  ])
  ring1 = [[10 * np.cos(a), 10 * np.sin(a), 0] for a in np.linspace(0, 2 * np.pi, 40)]
  c = Placement(base=(0, 25, 3), quat=quat((1, 0, 0), 40))
  ring2 = [c * np.array([6 * np.cos(a), 6 * np.sin(a), 0]) for a in np.linspace(0, 2 * np.pi, 40)]
  o, d = aimed_rays(20000, ring1 + ring2, 1.2, 1)
  g, gc, r, rc = run_both(tracer, oracle, sc, lim, o, d)
  assert rc['recorded_hits'] > 15000
  assert_same_short_paths(g, gc, r, rc, len(o))


def test_cylinder_cone_sphere_box(tracer, oracle):
  sc, lim = build([
      ('Lens', lambda d: [make.makeCylinder(d, 'Cy', 3, 8, base=(0, 0, 0), quat=quat((0, 1, 0), 30)),
                          make.makeCone(d, 'Co', 1, 4, 6, base=(12, 0, 0), quat=quat((1, 1, 0), 70))],
       dict(RefractiveIndex=1.7)),
      ('Mirror', lambda d: [make.makeSphere(d, 'Sp', 4, base=(-12, 3, 0)),
                            make.makeCone(d, 'Co2', 3, 0, 5, base=(0, -14, 0))], dict(Reflectivity=0.9)),
      ('Vacuum', lambda d: [make.makeBox(d, 'Bx', 6, 5, 4, base=(0, 12, -2), quat=quat((0, 0, 1), 25))], {}),
      ('Absorber', lambda d: [make.makeBox(d, 'Wall', 200, 200, 1, base=(-100, -100, -40))], {}),
  ])
  targets = [[0, 0, 3], [12, 2, 2], [-12, 3, 0], [0, -14, 2], [3, 14, 0]]
  o, d = aimed_rays(30000, targets, 2.5, 2)
  g, gc, r, rc = run_both(tracer, oracle, sc, lim, o, d)
  assert rc['recorded_hits'] > 40000
  assert_same(g, gc, r, rc)


def test_booleans(tracer, oracle):
  """Common, Cut (flipped tool normals), Fuse; nested Cut of a Common"""
  def lens(d):
    s = make.makeSphere(d, 'S', 8, base=(0, 0, -5))
    c = make.makeCylinder(d, 'C', 4, 10)
    return [make.makeCommon(d, [s, c], 'PlanoConvex', base=(0, 0, 10))]

  def drilled(d):
    b = make.makeBox(d, 'B', 10, 10, 10, base=(-5, -5, 0))
    h = make.makeCylinder(d, 'H', 2, 20, base=(0, 0, -5))
    return [make.makeCut(d, b, h, 'Drilled', base=(15, 0, 0))]

  def fused(d):
    a = make.makeSphere(d, 'A', 4, base=(0, 0, 0))
    b = make.makeSphere(d, 'Bs', 4, base=(5, 0, 0))
    return [make.makeFuse(d, [a, b], 'Peanut', base=(-18, 0, 5))]

  sc, lim = build([
      ('Lens', lens, dict(RefractiveIndex=1.5)),
      ('Mirror', drilled, {}),
      ('Lens', fused, dict(RefractiveIndex=1.3, name='OpticalLensGroup2')),
      ('Absorber', lambda d: [make.makeBox(d, 'Wall', 300, 300, 1, base=(-150, -150, -60))], {}),
  ])
  assert sc.n_prims == 7
  targets = [[0, 0, 12], [15, 0, 5], [-16, 0, 5]]
  o, d = aimed_rays(30000, targets, 2.0, 3)
  g, gc, r, rc = run_both(tracer, oracle, sc, lim, o, d)
  assert rc['recorded_hits'] > 40000
  assert_same(g, gc, r, rc)


def test_gratings_and_absorption(tracer, oracle):
  sc, lim = build([
      ('Grating', lambda d: [make.makeBox(d, 'G1', 20, 20, 2, base=(-10, -10, 20))],
       dict(GratingType='Reflection', GratingLinesPerMillimeter=600.0, GratingDiffractionOrder=1,
            GratingLinesOrientation=np.array([1.0, 0.0, 0.0]))),
      ('Grating', lambda d: [make.makeBox(d, 'G2', 20, 20, 2, base=(-10, -10, -30))],
       dict(GratingType='Transmission', GratingLinesPerMillimeter=300.0, GratingDiffractionOrder=-1,
            RefractiveIndex=1.5, GratingLinesOrientation=np.array([0.0, 1.0, 0.0]), name='OpticalGratingGroup2')),
      ('Lens', lambda d: [make.makeBox(d, 'Abs', 20, 20, 6, base=(30, -10, -3))],
       dict(RefractiveIndex=1.4, AbsorptionLength='3.0')),
      ('Absorber', lambda d: [make.makeSphere(d, 'Shell', 150)], {}),
  ])
  rs = np.random.RandomState(4)
  n = 6000
  o = np.zeros((n, 3))
  tg = np.array([[0, 0, 20], [0, 0, -29], [30, 0, 0]])[rs.randint(0, 3, n)] + rs.normal(0, 3, (n, 3))
  d = tg / np.linalg.norm(tg, axis=1)[:, None]
  g, gc, r, rc = run_both(tracer, oracle, sc, lim, o, d)
  assert rc['recorded_hits'] > 2 * n
  assert len(np.unique(r['power'])) > 100          # absorption really exercised
  assert_same(g, gc, r, rc)


def test_limits_and_sequence(tracer, oracle):
  sc, lim = build([
      ('Mirror', lambda d: [make.makeBox(d, 'A', 50, 50, 1, base=(-25, -25, 10)),
                            make.makeBox(d, 'B', 50, 50, 1, base=(-25, -25, -11))], dict(Reflectivity=0.7)),
  ], settings=dict(MaxIntersections=13.0, MaxRayLength=40.0))
  o, d = aimed_rays(5000, [[0, 0, 10], [0, 0, -10]], 8.0, 5, radius=3.0)
  g, gc, r, rc = run_both(tracer, oracle, sc, lim, o, d)
  assert rc['capped'] > 0 and rc['escaped'] > 0
  assert_same(g, gc, r, rc)
  # sequential mode with a per-ray sequence index
  sc2 = copy.copy(sc)
  sc2.seq_enabled = 1
  sc2.seq_mask = np.array([1, 1, 1], dtype=np.uint64)
  g, gc, r, rc = run_both(tracer, oracle, sc2, lim, o, d)
  assert rc['segments'] <= 4 * len(o)
  assert_same(g, gc, r, rc)


def test_stochastic_surfaces(tracer, oracle):
  """applyStochasticRayCorrections on the device (optical_group.py:279-323):
  a diffuse mirror (family over theta_in + azimuth dependence), a glossy lens
  with a ray modification, an ideal mirror in the same scene; every draw is
  keyed by (ray, intersection ordinal), so whole trajectories must agree"""
  sc, lim = build([
      ('Mirror', lambda d: [make.makeBox(d, 'M1', 30, 30, 1, base=(-15, -15, 20))],
       dict(ReflectedProbabilityDensity='cos(theta)**2*abs(sin(theta))*(1+0.5*cos(phi))*exp(-theta_in)',
            PowerThetaDomain='-pi, -pi/2', PowerPhiDomain='-pi, pi',
            RayModificationProbabilityDensity='exp(-theta**2/0.01)', ModifyThetaDomain='0, 0.3',
            ModifyPhiDomain='0, 2*pi')),
      ('Lens', lambda d: [make.makeSphere(d, 'L1', 6, base=(0, 0, -15))],
       dict(RefractiveIndex=1.5, RefractedProbabilityDensity='exp(-(theta-theta_refl)**2/0.002)',
            PowerThetaDomain='0, pi', PowerPhiDomain='-0.2, 0.2')),
      ('Mirror', lambda d: [make.makeBox(d, 'M2', 30, 1, 30, base=(-15, 25, -15))],
       dict(name='OpticalMirrorGroup2')),
      ('Absorber', lambda d: [make.makeBox(d, 'Wall', 400, 400, 1, base=(-200, -200, -60))], {}),
  ], settings=dict(MaxIntersections=12))
  assert [(s.group, s.kind, s.axis) for s in sc.surface_samplers] == [(0, 0, 1), (0, 1, 0), (1, 0, 2)]
  o, d = aimed_rays(20000, [[0, 0, 20], [0, 0, -15], [0, 25, 0]], 3.0, 11, radius=45.0)
  tracer.setSurfaceSeed(4242)
  tracer.setScene(sc)
  tracer.setLimits(lim)
  tracer.setDetector(None)
  tracer.reserveHits(len(o) * (lim.max_intersections + 1))
  tracer.reset()
  tracer.traceRays(o, d)
  tracer.sync()
  g, gc = tracer.hits(), tracer.counters()
  ref = oracle.trace_rays(sc, lim, o, d, surface_seed=4242, nthreads=8)
  r, rc = ref['hits'], ref['counters']
  assert rc['recorded_hits'] > 30000
  # the scattered directions differ from the ideal ones: a seed change moves the hits
  other = oracle.trace_rays(sc, lim, o[:2000], d[:2000], surface_seed=1, nthreads=8)['hits']
  assert not np.array_equal(other['tag'], r['tag'][:len(other)]) or \
      np.abs(other['point'] - r['point'][:len(other)]).max() > 1e-3
  assert_same_short_paths(g, gc, r, rc, len(o), max_len=12)
  tracer.setSurfaceSeed(0)


def test_stochastic_atoms_and_snell_families(tracer, oracle):
  """densities with DiracDelta terms (discrete events beside a continuum: a partly specular, partly diffuse mirror; a
  cone of fixed opening angle) and a lens density that names theta_in AND theta_refl (one table family per n1 / n2,
  picked by the hit): device == oracle on whole trajectories"""
  sc, lim = build([
      ('Mirror', lambda d: [make.makeBox(d, 'M1', 30, 30, 1, base=(-15, -15, 20))],
       dict(ReflectedProbabilityDensity='3*DiracDelta(theta-theta_refl)*DiracDelta(phi-phi_refl) + abs(cos(theta))*(1+0.3*cos(phi))',
            PowerThetaDomain='pi/2, pi', PowerPhiDomain='0, 2*pi')),
      ('Lens', lambda d: [make.makeSphere(d, 'L1', 6, base=(0, 0, -15))],
       dict(RefractiveIndex=1.5, RefractedProbabilityDensity='exp(-((theta-theta_refl)/(0.05+0.1*theta_in))**2)',
            PowerThetaDomain='0, pi', PowerPhiDomain='-0.3, 0.3')),
      ('Mirror', lambda d: [make.makeBox(d, 'M2', 30, 1, 30, base=(-15, 25, -15))],
       dict(name='OpticalMirrorGroup2', ReflectedProbabilityDensity='DiracDelta(theta-theta_refl) + 0.2*DiracDelta(theta-2.9)',
            PowerThetaDomain='pi/2, pi', PowerPhiDomain='0, 2*pi')),
      ('Absorber', lambda d: [make.makeBox(d, 'Wall', 400, 400, 1, base=(-200, -200, -60))], {}),
  ], settings=dict(MaxIntersections=12))
  kinds = [(s.group, s.kind, s.n_atoms, s.mu) for s in sc.surface_samplers]
  assert kinds[0] == (0, 0, 1, 0.0) and kinds[-1] == (2, 0, 2, 0.0)
  assert sorted(k[3] for k in kinds if k[0] == 1) == [-1.0, 1 / 1.5, 1.0, 1.5]      # vacuum and the lens itself as media
  o, d = aimed_rays(20000, [[0, 0, 20], [0, 0, -15], [0, 25, 0]], 3.0, 17, radius=45.0)
  tracer.setSurfaceSeed(99)
  tracer.setScene(sc)
  tracer.setLimits(lim)
  tracer.setDetector(None)
  tracer.reserveHits(len(o) * (lim.max_intersections + 1))
  tracer.reset()
  tracer.traceRays(o, d)
  tracer.sync()
  g, gc = tracer.hits(), tracer.counters()
  ref = oracle.trace_rays(sc, lim, o, d, surface_seed=99, nthreads=8)
  r, rc = ref['hits'], ref['counters']
  assert rc['recorded_hits'] > 30000
  assert_same_short_paths(g, gc, r, rc, len(o), max_len=12)
  tracer.setSurfaceSeed(0)


def test_stochastic_mirror_seeded_launch(tracer, oracle):
  """odw_trace (in-kernel generation) with a diffuse mirror: surface draws are
  keyed by the launch seed; counters, tags and histogram bins identical"""
  from freecad.optics_design_workbench_amd.freecad_elements import point_source
  doc = Document()
  make.makeMirror(doc, [make.makeBox(doc, length=20, width=20, height=1, base=(-10, -10, 10))],
                  ReflectedProbabilityDensity='cos(theta)**2 * abs(sin(theta))', PowerThetaDomain='-pi, -pi/2',
                  PowerPhiDomain='-pi, pi')
  make.makeAbsorber(doc, [make.makeBox(doc, length=400, width=400, height=1, base=(-200, -200, -50))])
  make.makeSimulationSettings(doc, MaxRayLength=1e4)
  src = make.makePointSource(doc, PowerDensity='exp(-theta**2/1e-2)')
  sc, lim, bs = bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)
  det = dict(group=1, origin=[0, 0, -49], ex=[1, 0, 0], ey=[0, 1, 0], x_lo=-200, x_hi=200, y_lo=-200, y_hi=200,
             nx=64, ny=64)
  n = 200000
  tracer.setScene(sc); tracer.setSource(bs); tracer.setLimits(lim); tracer.setDetector(det)
  tracer.reserveHits(n)
  tracer.reset()
  tracer.trace(5000, n, 77)
  tracer.sync()
  g, gc, gh = tracer.hits(), tracer.counters(), tracer.histogram()
  ref = oracle.trace(sc, bs, lim, 5000, n, 77, det=det, nthreads=8)
  assert gc == ref['counters']
  assert np.array_equal(g['tag'], ref['hits']['tag'])
  assert np.array_equal(gh, ref['hist'])
  assert np.abs(g['point'] - ref['hits']['point']).max() < 1e-7
  assert np.abs(g['direction'] - ref['hits']['direction']).max() < 1e-9
  # most of the diffusely reflected light lands on the big catcher
  assert gc['recorded_hits'] > 0.5 * n
  # re-uploading a scene without stochastic groups switches the samplers off again
  sc2 = bake.bakeScene(doc, src)
  sc2.surface_samplers = []
  tracer.setScene(sc2)
  tracer.reset()
  tracer.trace(5000, 1000, 77)
  tracer.sync()
  ref2 = oracle.trace(sc2, bs, lim, 5000, 1000, 77, det=det)
  assert tracer.counters() == ref2['counters']
  tracer.setDetector(None)


def test_box_edges_within_tolerance(tracer, oracle):
  """DistanceTolerance 1e-2 (test/50-old-tests/playground.FCStd): a ray that
  passes a box edge within the tolerance meets the widened rectangle of an
  exit face before that of an entry face; the nearest of ALL six faces counts"""
  sc, lim = build([
      ('Mirror', lambda d: [make.makeBox(d, 'B1', 10, 8, 1, base=(-5, -4, 0), quat=quat((1, 2, 0), 17))], {}),
      ('Lens', lambda d: [make.makeBox(d, 'B2', 6, 6, 6, base=(10, -3, -3), quat=quat((0, 1, 1), 31))],
       dict(RefractiveIndex=1.4)),
      ('Absorber', lambda d: [make.makeBox(d, 'Wall', 300, 300, 1, base=(-150, -150, -40))], {}),
  ], settings=dict(DistanceTolerance='1e-2', MaxIntersections=20))
  assert lim.dist_tol == 1e-2
  # targets on the edges and corners of both boxes
  targets = []
  for p in (0, 1):
    tw, size = sc.prim_to_world[p], sc.prim_params[p][:3]
    for a in np.linspace(0, 1, 9):
      for e in ((a, 0, 0), (a, 1, 0), (a, 0, 1), (a, 1, 1), (0, a, 0), (1, a, 0), (0, a, 1), (1, a, 1),
                (0, 0, a), (1, 0, a), (0, 1, a), (1, 1, a)):
        targets.append(tw * (np.array(e) * size))
  o, d = aimed_rays(60000, targets, 0.02, 5, radius=40.0)
  g, gc, r, rc = run_both(tracer, oracle, sc, lim, o, d)
  assert rc['recorded_hits'] > 40000
  assert_same_short_paths(g, gc, r, rc, len(o), max_len=20)
