"""Tessellated shapes (triangle-mesh / BVH fallback for faces that
are not quadrics): ODW_PRIM_TRIANGLE through bake, oracle and device.

The reference has no such path (OpenCASCADE intersects the exact surfaces), so
the pins are geometric: closed, outward-oriented tessellations; a tessellated
lens converges to the analytic lens as the facets shrink; the device equals
the oracle facet for facet."""
import os

import numpy as np
import pytest

from freecad.optics_design_workbench_amd.freecad_elements import make, point_source
from freecad.optics_design_workbench_amd.scene import Document, bake, geometry, stl


@pytest.mark.parametrize('kind, par, volume', [
  (geometry.SPHERE, (5, 0, 0, 0), 4 / 3 * np.pi * 125), (geometry.TORUS, (10, 2, 0, 0), 2 * np.pi**2 * 40),
  (geometry.CYLINDER, (2, 5, 0, 0), np.pi * 20), (geometry.CONE, (3, 1, 4, 0), np.pi * 4 / 3 * 13),
  (geometry.CONE, (3, 0, 4, 0), np.pi * 12), (geometry.BOX, (2, 3, 4, 0), 24.0)])
def test_tessellations_are_closed_and_outward(kind, par, volume):
  V, T, N = geometry.tessellate(kind, par, 96)
  p = V[T]
  fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
  vn = N[T].sum(axis=1)
  assert np.einsum('ij,ij->i', fn, vn).min() > 0                 # facets wound like the surface normals
  assert np.abs(np.linalg.norm(N, axis=1) - 1).max() < 1e-12
  vol = np.einsum('ij,ij->i', p[:, 0], np.cross(p[:, 1], p[:, 2])).sum() / 6   # divergence theorem: closed + outward
  assert abs(vol - volume) < 5e-3 * volume            # inscribed polyhedra: a few 1e-3 at 96 segments
  # every edge is shared by exactly two facets once the per-face vertices are welded
  Vw, Tw = stl.weld(V, T, tol=1e-9)
  e = np.sort(np.concatenate([Tw[:, [0, 1]], Tw[:, [1, 2]], Tw[:, [2, 0]]]), axis=1)
  _, counts = np.unique(e, axis=0, return_counts=True)
  assert np.all(counts == 2)


def test_stl_round_trip_and_crease_normals(tmp_path):
  V, T, N = geometry.tessellate(geometry.CYLINDER, (2, 5, 0, 0), 48)
  path = str(tmp_path / 'cyl.stl')
  stl.writeSTL(path, V, T)
  Vr, Tr = stl.readSTL(path)
  assert len(Tr) == len(T) and len(Vr) < len(V)                  # welded
  open(str(tmp_path / 'a.stl'), 'w').write(
      'solid a\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid a\n')
  va, ta = stl.readSTL(str(tmp_path / 'a.stl'))
  assert va.shape == (3, 3) and va[ta].tolist() == [[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]   # winding kept
  Vs, Ts, Ns = stl.smoothNormals(Vr, Tr, creaseAngleDeg=30)
  # lateral vertices get radial normals, cap vertices axial ones: the rim stays sharp
  rad = np.hypot(Vs[:, 0], Vs[:, 1])
  axial = np.abs(Ns[:, 2]) > 0.99
  lateral = ~axial
  assert axial.sum() > 10 and lateral.sum() > 10
  radial = np.stack([Vs[lateral, 0], Vs[lateral, 1]], 1) / rad[lateral, None]
  assert np.abs(np.einsum('ij,ij->i', radial, Ns[lateral, :2]) - 1).max() < 1e-3     # float32 file
  with pytest.raises(ValueError):
    open(str(tmp_path / 'bad.stl'), 'w').write('hello')
    stl.readSTL(str(tmp_path / 'bad.stl'))


def _lens_scene(segments=None, smooth=True, n_src='exp(-theta**2/0.05**2)'):
  doc = Document()
  sp = make.makeSphere(doc, 'S', 5, base=(0, 0, 30))
  el = [sp] if segments is None else [make.makeTessellated(doc, sp, segments, smooth=smooth)]
  make.makeLens(doc, el, RefractiveIndex=1.5)
  make.makeAbsorber(doc, [make.makeBox(doc, 'A', 100, 100, 1, base=(-50, -50, 60))])
  make.makeSimulationSettings(doc)
  src = make.makePointSource(doc, PowerDensity=n_src)
  return doc, bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)


def test_convex_tessellated_solids_are_recognised(oracle):
  """`geometry.mesh_is_convex`: closed, outward, every edge convex -- a tessellated ball, box, cylinder and cone are
  convex polyhedra, a torus is not, an open patch is not, an inside-out ball is not.  The bake flags their facets
  ODW_FLAG_CONVEX, and with the flag the tracer drops the solid's facets for the segment after a ray has left it
  (decided with the facet's own normal): the same rows as without the rule (reference-strict mode of the oracle)."""
  for kind, params, convex in ((geometry.SPHERE, (5.0, 0, 0, 0), True), (geometry.BOX, (1.0, 2.0, 3.0, 0), True),
                               (geometry.CYLINDER, (2.0, 5.0, 0, 0), True), (geometry.CONE, (2.0, 1.0, 5.0, 0), True),
                               (geometry.TORUS, (5.0, 1.0, 0, 0), False)):
    v, tri, _ = geometry.tessellate(kind, params, 24)
    assert geometry.mesh_is_convex(v, tri) is convex
    # (2: no edge bends outward by more than rounding -- ODW_FLAG_STRICTLY_CONVEX, what the mesh kernel's normal cones need)
    assert geometry.mesh_convexity(v, tri) == (2 if convex else 0)
  v, tri, _ = geometry.tessellate(geometry.SPHERE, (5.0, 0, 0, 0), 24)
  # a vertex pulled OUT by 1e-10 of its radius: the diagonal of a (planar) quadrilateral of the tessellation now bends
  # outward by more than rounding, within the tolerance of the test -- convex for the exit rule, not strictly; by 1e-6: not convex
  k = int(np.argmax(v[:, 0]))
  ring = np.linalg.norm(v - v[k], axis=1) < 1e-9
  for eps, level in ((1e-10, 1), (1e-6, 0)):
    w = v.copy()
    w[ring] = v[ring] * (1 + eps)
    assert geometry.mesh_convexity(w, tri) == level
  assert not geometry.mesh_is_convex(v, tri[:-3]) and not geometry.mesh_is_convex(v, tri[:, ::-1])
  dented = v.copy()
  k = int(np.argmax(v[:, 0]))
  dented[np.linalg.norm(v - v[k], axis=1) < 1e-9] *= 0.8          # one vertex (and its seam twins) pushed inwards
  assert not geometry.mesh_is_convex(dented, tri)
  _, sc, lim, src = _lens_scene(48)
  tri_rows = sc.prim_type == geometry.TRIANGLE
  assert np.all(sc.prim_flags[tri_rows] & 2) and np.all(sc.prim_flags[tri_rows] & 8) and sc.prim_flags[~tri_rows].tolist() == [2 | (63 << 8)]
  n = 200000
  a = oracle.trace(sc, src, lim, 0, n, 3, nthreads=0)
  with oracle.strict():
    b = oracle.trace(sc, src, lim, 0, n, 3, nthreads=0)
  assert a['counters'] == b['counters'] and a['counters']['recorded_hits'] > 0.9 * n
  assert np.array_equal(a['hits']['tag'], b['hits']['tag']) and np.array_equal(a['hits']['point'], b['hits']['point'])


def test_bake_of_meshes():
  _, sc, _, _ = _lens_scene(32)
  tri = sc.prim_type == geometry.TRIANGLE
  assert tri.sum() == sc.n_prims - 1 and sc.prim_type[0] == geometry.BOX        # analytic primitives first
  assert sc.tri_normals.shape == (sc.n_prims, 9) and np.all(sc.tri_normals[0] == 0)
  v = sc.prim_xform[tri][:, :9].reshape(-1, 3, 3)
  assert np.abs(np.linalg.norm(v - [0, 0, 30], axis=2) - 5).max() < 1e-12        # placed vertices on the sphere
  assert np.all(sc.prim_cond_off == 0) and np.all(sc.prim_group[tri] == 0)
  assert sc.tri_normals is not None and _lens_scene(32, smooth=False)[1].tri_normals is None
  doc = Document()
  m = make.makeTessellated(doc, make.makeSphere(doc, 'S', 5), 16)
  cut = make.makeCut(doc, make.makeBox(doc, 'B'), m)
  make.makeLens(doc, [cut])
  make.makeSimulationSettings(doc)
  with pytest.raises(geometry.UnsupportedGeometry):
    bake.bakeScene(doc, make.makePointSource(doc))


def test_tessellated_lens_converges_to_the_analytic_lens(oracle):
  """ball lens, same rays: hit positions on the screen approach the analytic
  ones like the chord error (~1/segments^2) with interpolated normals; facet
  normals alone leave an error of the order of the facet angle"""
  n = 3000
  _, a, lim, src = _lens_scene()
  exact = oracle.trace(a, src, lim, 0, n, 1, nthreads=4)
  ref = exact['hits']['point']
  err = {}
  for seg in (24, 48, 96):
    _, m, lim, src = _lens_scene(seg)
    r = oracle.trace(m, src, lim, 0, n, 1, nthreads=4)
    assert r['counters'] == exact['counters']
    err[seg] = np.sqrt(np.mean(np.sum((r['hits']['point'] - ref)**2, axis=1)))
  assert err[48] < 0.45 * err[24] and err[96] < 0.45 * err[48] and err[96] < 0.05
  _, flat, lim, src = _lens_scene(96, smooth=False)
  flat_err = np.sqrt(np.mean(np.sum((oracle.trace(flat, src, lim, 0, n, 1, nthreads=4)['hits']['point'] - ref)**2, axis=1)))
  assert flat_err > 5 * err[96]


# ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def tracer(native_lib):
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  tr = Tracer(0)
  yield tr
  tr.close()


def _mixed_scene():
  """tessellated torus lens + tessellated (facet-normal) cone mirror + analytic sphere lens + wall"""
  doc = Document()
  to = make.makeTorus(doc, 'T', 8, 2.5, base=(0, 0, 20))
  co = make.makeCone(doc, 'C', 4, 1, 6, base=(14, 0, 18), quat=(np.sin(0.3), 0, 0, np.cos(0.3)))
  make.makeLens(doc, [make.makeTessellated(doc, to, 40)], RefractiveIndex=1.4, RecordHits=True)
  make.makeMirror(doc, [make.makeTessellated(doc, co, 24, smooth=False)], RecordHits=True)
  make.makeLens(doc, [make.makeSphere(doc, 'S', 3, base=(-12, 0, 20))], RefractiveIndex=1.7, RecordHits=True,
                name='OpticalLensGroup2')
  make.makeAbsorber(doc, [make.makeBox(doc, 'W', 300, 300, 1, base=(-150, -150, 60))])
  make.makeSimulationSettings(doc, MaxIntersections=30.0)
  src = make.makePointSource(doc, PowerDensity='1', ThetaDomain='0, 0.9')
  return bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)


@pytest.mark.gpu
def test_device_equals_oracle_on_meshes(tracer, oracle):
  sc, lim, src = _mixed_scene()
  assert (sc.prim_type == geometry.TRIANGLE).sum() > 1500
  n = 20000
  tracer.setScene(sc); tracer.setSource(src); tracer.setLimits(lim); tracer.setDetector(None)
  tracer.reserveHits(n * 31)
  tracer.reset()
  tracer.trace(0, n, 7)
  tracer.sync()
  g, gc = tracer.hits(), tracer.counters()
  ref = oracle.trace(sc, src, lim, 0, n, 7, hit_capacity=n * 31, nthreads=8)
  r, rc = ref['hits'], ref['counters']
  assert rc['recorded_hits'] > 1.5 * n
  # rays trapped inside the torus lens bounce many times and amplify rounding: compare short paths exactly
  m48 = np.uint64(0xFFFFFFFFFFFF)
  gr, rr = (g['tag'] & m48).astype(np.int64), (r['tag'] & m48).astype(np.int64)
  cg, cr = np.bincount(gr, minlength=n), np.bincount(rr, minlength=n)
  ok = (cg == cr) & (cg <= 8)
  assert ok.mean() > 0.97 and (cg != cr).mean() < 5e-3
  sg, sr = ok[gr], ok[rr]
  assert np.array_equal(g['tag'][sg], r['tag'][sr])
  dp = np.abs(g['point'][sg] - r['point'][sr]).max(axis=1)
  assert (dp < 1e-7).mean() > 0.999 and dp.max() < 1e-3
  assert gc['traced_rays'] == rc['traced_rays'] == n


@pytest.mark.gpu
def test_every_segment_of_the_mesh_kernel_against_the_oracle(tracer, oracle):
  """rays trapped inside the torus lens amplify rounding, whole trajectories are compared for short paths only (above);
  single segments need no such allowance: hit k >= 1 of a ray ends the segment that starts at hit k - 1 with the direction
  row k carries.  The oracle traces that segment alone (explicit ray, one intersection) and must land on the device's hit
  -- same group and side, the point within 1e-9 mm -- for every recorded segment of the mixed scene (facets with
  interpolated and with facet normals, an analytic sphere, distTol 1e-6) and of a ball of 6.5e4 facets at distTol 1e-2
  (normal cones armed for the segments inside the ball)."""
  import copy
  cases = [(_mixed_scene(), 20000, 30)]
  _, sc2, lim2, src2 = _lens_scene(256)
  lim2 = copy.copy(lim2)
  lim2.dist_tol = 1e-2
  cases.append(((sc2, lim2, src2), 50000, 8))
  for (sc, lim, src), n, per in cases:
    sc = copy.copy(sc)
    sc.group_record = np.ones_like(sc.group_record)
    tracer.setScene(sc); tracer.setSource(src); tracer.setLimits(lim); tracer.setDetector(None)
    tracer.reserveHits(n * (per + 1))
    tracer.reset()
    tracer.trace(0, n, 23)
    tracer.sync()
    g = tracer.hits()
    assert tracer.counters()['hits_dropped'] == 0
    m48 = np.uint64(0xFFFFFFFFFFFF)
    ray = (g['tag'] & m48).astype(np.int64)
    later = np.nonzero(ray[1:] == ray[:-1])[0] + 1
    assert len(later) > n // 2
    one = copy.copy(lim)
    one.max_intersections = 1
    o = oracle.trace_rays(sc, one, g['point'][later - 1], g['direction'][later], flags=1, nthreads=0)['hits']
    assert len(o) == len(later) and np.array_equal((o['tag'] & m48).astype(np.int64), np.arange(len(later)))
    assert np.array_equal(o['tag'] >> np.uint64(48), g['tag'][later] >> np.uint64(48))
    assert np.abs(o['point'] - g['point'][later]).max() < 1e-9


@pytest.mark.gpu
def test_mesh_kernel_equals_bvh_kernel(native_lib, monkeypatch):
  """launches without stochastic surfaces and segment rows take odw_mesh_kernel (node / leaf state machine, float32
  leaf filter in front of the float64 facet test); ODW_MESH_KERNEL=0 keeps odw_trace_kernel<true, ...>.  The filter
  only discards facets the float64 test would discard: the same rows, bit for bit, on the mixed scene (facets +
  analytic primitives in one tree, distTol 1e-6) and on a ball of 6.5e4 facets with distTol 1e-2 (wide rims)."""
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  cases = [(_mixed_scene(), 30000)]
  _, sc, lim, src = _lens_scene(256)
  cases.append(((sc, lim, src), 200000))
  doc2, sc2, lim2, src2 = _lens_scene(64)
  import copy
  lim2 = copy.copy(lim2)
  lim2.dist_tol = 1e-2
  cases.append(((sc2, lim2, src2), 100000))
  # stochastic surfaces (odw_mesh_kernel<true>): a tessellated ball with a scattering refraction in front of a
  # tessellated diffuse mirror
  doc3 = Document()
  ball = make.makeTessellated(doc3, make.makeSphere(doc3, 'S', 5, base=(0, 0, 30)), 48)
  make.makeLens(doc3, [ball], RefractiveIndex=1.5, RefractedProbabilityDensity='exp(-(theta-theta_refl)**2/0.005)')
  dish = make.makeTessellated(doc3, make.makeSphere(doc3, 'M', 40, base=(0, 0, 100)), 64)
  make.makeMirror(doc3, [dish], ReflectedProbabilityDensity='cos(theta-theta_refl)**8', RecordHits=True)
  make.makeAbsorber(doc3, [make.makeBox(doc3, 'A', 200, 200, 1, base=(-100, -100, -20))])
  make.makeSimulationSettings(doc3, MaxIntersections=12.0)
  src3 = make.makePointSource(doc3, PowerDensity='exp(-theta**2/0.1**2)')
  cases.append(((bake.bakeScene(doc3, src3), bake.bakeLimits(doc3, src3), point_source.bakeSource(doc3, src3)), 50000))
  for (sc, lim, src), n in cases:
    rows = {}
    for mode in ('1', '0'):
      monkeypatch.setenv('ODW_MESH_KERNEL', mode)
      with Tracer(0) as tr:
        tr.setScene(sc); tr.setSource(src); tr.setLimits(lim); tr.setDetector(None)
        tr.reserveHits(n * 8)
        tr.reset()
        tr.trace(0, n, 11, histogram=False)
        tr.sync()
        rows[mode] = (tr.counters(), tr.hits())
    assert rows['1'][0] == rows['0'][0] and rows['1'][0]['recorded_hits'] > n // 2
    a, b = rows['1'][1], rows['0'][1]
    assert np.array_equal(a['tag'], b['tag'])
    assert np.array_equal(a['point'], b['point']) and np.array_equal(a['direction'], b['direction'])
    assert np.array_equal(a['power'], b['power'])


@pytest.mark.gpu
def test_normal_cones_change_no_row(native_lib, monkeypatch):
  """a ray that travels inside a convex tessellated solid drops the slots of the eight-wide tree whose facets all face
  it (WideBvh::cone_word: the neighbourhood of the facet it starts on) -- facets it could only meet from outside, at a
  distance below distTol.  The rows are those without the cones (ODW_MESH_CONES=0) and those of the binary kernel, bit
  for bit: a coarse ball under a beam that covers it up to grazing incidence (wide cones, long facets), a fine one, a
  cylinder and a cone as lenses (several faces: rays that enter next to a rim start on a facet with an open edge and get
  no cones), a ball mirror that rays never enter, total reflection inside a ball of high index."""
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  def scene(solid, n_seg, index=1.5, theta='0, 0.25', mirror=False, max_int=30.0, **kw):
    doc = Document()
    shape = make.makeTessellated(doc, solid(doc), n_seg, **kw)
    if mirror:
      make.makeMirror(doc, [shape], RecordHits=True)
    else:
      make.makeLens(doc, [shape], RefractiveIndex=index, RecordHits=True)
    make.makeAbsorber(doc, [make.makeBox(doc, 'A', 400, 400, 1, base=(-200, -200, 80))])
    make.makeSimulationSettings(doc, MaxIntersections=max_int)
    src = make.makePointSource(doc, PowerDensity='1', ThetaDomain=theta)
    return bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)
  ball = lambda doc: make.makeSphere(doc, 'S', 5, base=(0, 0, 25))
  cases = [(scene(ball, 12), 400_000), (scene(ball, 160), 400_000), (scene(ball, 48, index=2.6), 300_000),
           (scene(lambda doc: make.makeCylinder(doc, 'C', 4, 9, base=(0, -1, 22), quat=(np.sin(0.4), 0, 0, np.cos(0.4))), 40, smooth=False), 300_000),
           (scene(lambda doc: make.makeCone(doc, 'K', 5, 1.5, 7, base=(0.5, 0, 22), quat=(0, np.sin(0.5), 0, np.cos(0.5))), 32, smooth=False), 300_000),
           (scene(ball, 32, mirror=True), 200_000)]
  for (sc, lim, src), n in cases:
    assert (sc.prim_flags[sc.prim_type == geometry.TRIANGLE] & 2).all()          # (convex: the rule is armed)
    rows = {}
    for mode, kernel, cones in (('cones', '1', '1'), ('plain', '1', '0'), ('binary', '0', '1')):
      monkeypatch.setenv('ODW_MESH_KERNEL', kernel)
      monkeypatch.setenv('ODW_MESH_CONES', cones)
      with Tracer(0) as tr:
        tr.setScene(sc); tr.setSource(src); tr.setLimits(lim); tr.setDetector(None)
        tr.reserveHits(n * 6)
        tr.reset()
        tr.trace(3, n, 17, histogram=False)
        tr.sync()
        rows[mode] = (tr.counters(), tr.hits())
    assert rows['cones'][0]['hits_dropped'] == 0 and rows['cones'][0]['recorded_hits'] > n // 4
    for other in ('plain', 'binary'):
      assert rows['cones'][0] == rows[other][0], other
      for col in ('tag', 'point', 'direction', 'power'):
        assert np.array_equal(rows['cones'][1][col], rows[other][1][col]), (other, col)


@pytest.mark.gpu
def test_presorted_hand_out_order_changes_no_row(native_lib, monkeypatch):
  """mesh launches hand their rays out sorted by where they start and point (TraceParams.ray_order: the rays of a wave
  are neighbours, their fetches share cache lines); a ray's rows depend on its number only -- the same rows with the
  order (forced for short launches: ODW_MESH_PRESORT_MIN=1) and without it (ODW_MESH_PRESORT=0), for a point source
  at its focus (direction bits only), a diverging one and a parallel beam (origin bits), first ray not 0"""
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  import copy
  _, sc, lim, src = _lens_scene(96)
  far, par = copy.copy(src), copy.copy(src)
  far.focal_length, par.focal_length = -40.0, np.inf
  for source, n, first in ((src, 70001, 12345), (far, 30000, 0), (par, 30000, 7)):
    rows = {}
    for mode in ('on', 'off'):
      monkeypatch.setenv('ODW_MESH_PRESORT', '1' if mode == 'on' else '0')
      monkeypatch.setenv('ODW_MESH_PRESORT_MIN', '1')
      with Tracer(0) as tr:
        tr.setScene(sc); tr.setSource(source); tr.setLimits(lim); tr.setDetector(None)
        tr.reserveHits(n * 8)
        tr.reset()
        tr.trace(first, n, 5, histogram=False)
        tr.sync()
        rows[mode] = (tr.counters(), tr.hits())
    assert rows['on'][0] == rows['off'][0] and rows['on'][0]['traced_rays'] == n
    for col in ('tag', 'point', 'direction', 'power'):
      assert np.array_equal(rows['on'][1][col], rows['off'][1][col]), col


@pytest.mark.gpu
def test_mesh_kernel_on_a_lopsided_tree(native_lib, oracle, monkeypatch):
  """facets whose sizes and spacings grow geometrically along an axis (a horn of 1200 quadrilateral rings between
  1e-3 and 1e3 mm): the surface-area heuristic peels them off one side, the binary tree comes out as high as the
  builder lets it, and the eight-wide tree of the mesh kernel has to open the tallest subtrees first to stay inside
  its 12 levels -- or the library falls back to the binary kernels.  Either way: the rows of the oracle."""
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  rings, sides = 1200, 6
  z = 1e-3 * (1e6 ** (np.arange(rings + 1) / rings))
  r = 0.3 * z**0.9                      # (a ray from the apex at angle t meets the wall where 0.3 z^-0.1 = tan t)
  ang = 2 * np.pi * np.arange(sides) / sides
  v = np.concatenate([np.column_stack([rk * np.cos(ang), rk * np.sin(ang), np.full(sides, zk)]) for rk, zk in zip(r, z)])
  tri = []
  for k in range(rings):
    for j in range(sides):
      a, b = k * sides + j, k * sides + (j + 1) % sides
      c, d = a + sides, b + sides
      tri += [[a, b, d], [a, d, c]]
  doc = Document()
  make.makeMirror(doc, [make.makeMesh(doc, v, np.array(tri), placement=None)], RecordHits=True)
  make.makeAbsorber(doc, [make.makeBox(doc, 'A', 4000, 4000, 1, base=(-2000, -2000, 1500))])
  make.makeSimulationSettings(doc, MaxIntersections=40.0)
  src = make.makePointSource(doc, PowerDensity='1', ThetaDomain='0.1, 0.62')
  sc, lim, bs = bake.bakeScene(doc, src), bake.bakeLimits(doc, src), point_source.bakeSource(doc, src)
  assert (sc.prim_type == geometry.TRIANGLE).sum() == 2 * rings * sides
  n = 20000
  ref = oracle.trace(sc, bs, lim, 0, n, 5, flags=1, nthreads=0, hit_capacity=n * 41)
  rows = {}
  for mode in ('1', '0'):
    monkeypatch.setenv('ODW_MESH_KERNEL', mode)
    with Tracer(0) as tr:
      tr.setScene(sc); tr.setSource(bs); tr.setLimits(lim); tr.setDetector(None)
      tr.reserveHits(n * 41)
      tr.reset()
      tr.trace(0, n, 5, histogram=False)
      tr.sync()
      rows[mode] = (tr.counters(), tr.hits())
  assert rows['1'][0] == rows['0'][0]
  assert np.array_equal(rows['1'][1]['tag'], rows['0'][1]['tag']) and np.array_equal(rows['1'][1]['point'], rows['0'][1]['point'])
  # against the oracle: the source sits at the horn's apex, rays meet the wall between 1e-3 and 1e3 mm and bounce on
  c, h = rows['1']
  assert c['recorded_hits'] > 2 * n
  m48 = np.uint64(0xFFFFFFFFFFFF)
  gr, rr = (h['tag'] & m48).astype(np.int64), (ref['hits']['tag'] & m48).astype(np.int64)
  cg, cr = np.bincount(gr, minlength=n), np.bincount(rr, minlength=n)
  ok = (cg == cr) & (cg <= 10)
  assert (cg == cr).mean() > 0.995 and ok.mean() > 0.4
  assert np.array_equal(h['tag'][ok[gr]], ref['hits']['tag'][ok[rr]])
  assert np.abs(h['point'][ok[gr]] - ref['hits']['point'][ok[rr]]).max() < 1e-6


@pytest.mark.gpu
def test_large_mesh_matches_analytic_statistics(tracer):
  """2e5 facets (BVH with binned SAH): the ball-lens spot of the tessellated
  lens equals the analytic one within the facet error, all rays accounted for"""
  n = 400000
  res = {}
  for seg in (None, 448):
    _, sc, lim, src = _lens_scene(seg)
    tracer.setScene(sc); tracer.setSource(src); tracer.setLimits(lim); tracer.setDetector(None)
    tracer.reserveHits(n + 1024)
    tracer.reset()
    tracer.trace(0, n, 3)
    tracer.sync()
    c, h = tracer.counters(), tracer.hits()
    assert c['traced_rays'] == n and c['recorded_hits'] + c['escaped'] + c['capped'] >= n - 5
    res[seg] = (c, h['point'][np.argsort((h['tag'] & np.uint64(0xFFFFFFFFFFFF)))])
  assert sc.n_prims > 190000
  ca, cm = res[None][0], res[448][0]
  assert ca['recorded_hits'] == cm['recorded_hits'] and abs(ca['segments'] - cm['segments']) < 20
  assert np.sqrt(np.mean(np.sum((res[None][1] - res[448][1])**2, axis=1))) < 5e-3


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_single_leaf_tree_terminates(native_lib, oracle):
  """a mesh of one facet (and one of four overlapping ones) makes a tree that is a single leaf:
  the root is a wrapper whose second child does not exist.  Round 1 found its inverted box being
  met by every ray, which sent the traversal back to the root for ever (randomised parity run with
  the BVH kernels forced)."""
  from freecad.optics_design_workbench_amd.freecad_elements import make
  from freecad.optics_design_workbench_amd.scene import Document, bake
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
  for tris in ([[0, 1, 2]], [[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]]):
    doc = Document()
    v = np.array([[-5, -5, 20.0], [5, -5, 20.0], [0, 5, 20.0], [0, 0, 24.0]])
    make.makeMirror(doc, [make.makeMesh(doc, v, np.array(tris))], RecordHits=True)
    make.makeSimulationSettings(doc)
    src = make.makePointSource(doc, PowerDensity='1', ThetaDomain='0, 0.3')
    from freecad.optics_design_workbench_amd.simulation.simulation_loop import bakeLightSource
    sc, bs, lim = bake.bakeScene(doc, src), bakeLightSource(doc, src, 0), bake.bakeLimits(doc, src)
    n = 20000
    with Tracer(0) as tr:
      tr.setScene(sc); tr.setSource(bs); tr.setLimits(lim); tr.setDetector(None)
      tr.reserveHits(4 * n)
      tr.reset()
      tr.trace(0, n, 1, histogram=False)
      tr.sync()
      ref = oracle.trace(sc, bs, lim, 0, n, 1, flags=1, nthreads=0)
      assert tr.counters() == ref['counters'] and ref['counters']['recorded_hits'] > 1000
      assert np.array_equal(tr.hits()['tag'], ref['hits']['tag'])


def test_tolerance_widens_facets_across_face_edges_only(oracle):
  """`dist(point, face) < distTol` (ray.py:424-426) is about faces: a facet is widened across the
  edges it has in common with the face's outline, not across those it shares with a neighbouring
  facet of the same face.  With distTol = 1e-2 -- larger than the needles of a pole fan -- a
  closed tessellated ball then gives the same hits as with 1e-6, and an open patch still catches
  rays that pass within distTol of its rim."""
  from freecad.optics_design_workbench_amd.freecad_elements import make
  from freecad.optics_design_workbench_amd.scene import Document, bake
  from freecad.optics_design_workbench_amd.scene.bake import faceEdgeBits
  quad = np.array([[0, 1, 2], [0, 2, 3]])
  assert faceEdgeBits(quad).tolist() == [6, 5]            # the diagonal (v0-v2) is interior in both
  assert faceEdgeBits(np.array([[0, 1, 2]])).tolist() == [7]

  def hits(tol, elements, o, d):
    doc = Document()
    make.makeMirror(doc, [e(doc) for e in elements], RecordHits=True)
    make.makeSimulationSettings(doc, DistanceTolerance=str(tol), MaxIntersections=1.0)
    src = make.makePointSource(doc)
    sc, lim = bake.bakeScene(doc, src), bake.bakeLimits(doc, src)
    return sc, oracle.trace_rays(sc, lim, o, d, flags=1)['hits']
  rs = np.random.RandomState(2)
  o = rs.normal(0, 1, (4000, 3)); o = o / np.linalg.norm(o, axis=1)[:, None] * 30
  d = rs.normal(0, 1.5, (4000, 3)) - o; d /= np.linalg.norm(d, axis=1)[:, None]
  ball = [lambda doc: make.makeTessellated(doc, make.makeSphere(doc, 'S', 5), 12)]
  sc, fine = hits(1e-6, ball, o, d)
  assert sc.tri_edges is not None and (sc.tri_edges[sc.prim_type == 5] != 7).all()      # a closed surface
  _, coarse = hits(1e-2, ball, o, d)
  assert np.array_equal(fine['tag'], coarse['tag']) and np.abs(fine['point'] - coarse['point']).max() < 1e-9
  # an open patch: one facet; rays aimed just outside its rim
  patch = [lambda doc: make.makeMesh(doc, np.array([[-1, -1, 0.0], [1, -1, 0], [0, 1, 0]]), np.array([[0, 1, 2]]))]
  o2 = np.array([[0.0, -1.005, 5.0], [0.0, -1.02, 5.0], [0.0, -0.5, 5.0]])
  d2 = np.tile([0.0, 0.0, -1.0], (3, 1))
  _, h = hits(1e-2, patch, o2, d2)
  assert [int(t) & 0xFFFFFFFF for t in h['tag']] == [0, 2]                 # 5e-3 outside: caught; 2e-2 outside: missed
  _, h = hits(1e-6, patch, o2, d2)
  assert [int(t) & 0xFFFFFFFF for t in h['tag']] == [2]
