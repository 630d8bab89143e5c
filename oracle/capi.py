"""ctypes binding of the CPU oracle (oracle/libodw_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.

The structures restate include/odw_trace.h; inputs are the product's baked
tables (duck-typed: any object with the same numpy array attributes).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libodw_oracle.so')

CNT_NAMES = ['traced_rays', 'recorded_hits', 'segments', 'escaped', 'died', 'capped',
             'hist_overflow', 'hits_dropped', 'grating_in_medium']
TRACE_RECORD_HITS, TRACE_HISTOGRAM = 1, 2

HIT_DTYPE = np.dtype([('point', '<f8', 3), ('direction', '<f8', 3), ('power', '<f8'), ('tag', '<u8')])
SEGMENT_DTYPE = np.dtype([('p1', '<f8', 3), ('p2', '<f8', 3), ('power', '<f8'), ('tag', '<u8')])

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu = C.POINTER(C.c_uint64)


class SceneDesc(C.Structure):
  _fields_ = [('n_prims', C.c_int32), ('prim_type', _pi), ('prim_group', _pi), ('prim_solid', _pi),
              ('prim_flags', _pi), ('prim_xform', _pd), ('prim_params', _pd), ('prim_cond_off', _pi),
              ('n_conds', C.c_int32), ('cond_prim', _pi), ('cond_inside', _pi),
              ('n_groups', C.c_int32), ('group_type', _pi), ('group_ior', _pd), ('group_refl', _pd),
              ('group_abslen', _pd), ('group_record', _pi), ('group_grating_type', _pi),
              ('group_grating_lpm', _pd), ('group_grating_dir', _pd), ('group_grating_order', _pi),
              ('seq_enabled', C.c_int32), ('seq_len', C.c_int32), ('seq_mask', _pu),
              ('ignore_mask', C.c_uint64), ('tri_normals', _pd), ('tri_edges', _pi)]


class SourceDesc(C.Structure):
  _fields_ = [('xform', C.c_double * 12), ('focal_length', C.c_double), ('wavelength', C.c_double),
              ('power', C.c_double), ('n_phi_knots', C.c_int32), ('phi_edges', _pd), ('phi_cdf', _pd),
              ('n_t_knots', C.c_int32), ('n_t_rows', C.c_int32), ('t_edges', _pd), ('t_cdf', _pd)]


class LimitsDesc(C.Structure):
  _fields_ = [('max_ray_length', C.c_double), ('max_intersections', C.c_int32),
              ('dist_tol', C.c_double), ('power_tol', C.c_double)]


class DetectorDesc(C.Structure):
  _fields_ = [('group', C.c_int32), ('origin', C.c_double * 3), ('ex', C.c_double * 3),
              ('ey', C.c_double * 3), ('x_lo', C.c_double), ('x_hi', C.c_double),
              ('y_lo', C.c_double), ('y_hi', C.c_double), ('nx', C.c_int32), ('ny', C.c_int32)]


class SurfaceSourceDesc(C.Structure):
  _fields_ = [('wavelength', C.c_double), ('power', C.c_double), ('dist_tol', C.c_double),
              ('n_prims', C.c_int32), ('prim_type', _pi), ('prim_flags', _pi), ('prim_xform', _pd),
              ('prim_params', _pd), ('prim_cond_off', _pi), ('n_conds', C.c_int32), ('cond_prim', _pi),
              ('cond_inside', _pi), ('n_faces', C.c_int32), ('face_prim', _pi), ('face_id', _pi),
              ('face_area', _pd), ('n_t_knots', C.c_int32), ('t_edges', _pd), ('t_cdf', _pd), ('tri_normals', _pd)]


class SurfaceSamplerDesc(C.Structure):
  _fields_ = [('group', C.c_int32), ('kind', C.c_int32), ('family_axis', C.c_int32), ('n_family', C.c_int32),
              ('family_lo', C.c_double), ('family_hi', C.c_double), ('n_phi_knots', C.c_int32),
              ('phi_edges', _pd), ('phi_cdf', _pd), ('n_t_knots', C.c_int32), ('n_t_rows', C.c_int32),
              ('t_edges', _pd), ('t_cdf', _pd), ('mu', C.c_double), ('n_atoms', C.c_int32),
              ('atom_mass', _pd), ('atom_theta', _pd), ('atom_phi', _pd)]


def build(force=False):
  if force or not os.path.exists(_LIB_PATH) or \
      os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, 'odw_oracle.c')):
    subprocess.check_call(['make', '-C', _HERE, 'libodw_oracle.so'], stdout=subprocess.DEVNULL)
  return _LIB_PATH


_lib = None


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = C.CDLL(_LIB_PATH)
    _lib.odw_oracle_interp.restype = C.c_double
    _lib.odw_oracle_interp.argtypes = [C.c_double, _pd, _pd, C.c_int32]
  return _lib


def _arr(a, dtype):
  return np.ascontiguousarray(a, dtype=dtype)


def _p(a, typ):
  return a.ctypes.data_as(typ)


class _Keep:
  """desc + the arrays it points to"""

  def __init__(self, desc, keep):
    self.desc, self.keep = desc, keep


def scene_desc(sc):
  k = dict(
      prim_type=_arr(sc.prim_type, np.int32), prim_group=_arr(sc.prim_group, np.int32),
      prim_solid=_arr(sc.prim_solid, np.int32), prim_flags=_arr(sc.prim_flags, np.int32),
      prim_xform=_arr(sc.prim_xform, np.float64), prim_params=_arr(sc.prim_params, np.float64),
      prim_cond_off=_arr(sc.prim_cond_off, np.int32), cond_prim=_arr(sc.cond_prim, np.int32),
      cond_inside=_arr(sc.cond_inside, np.int32), group_type=_arr(sc.group_type, np.int32),
      group_ior=_arr(sc.group_ior, np.float64), group_refl=_arr(sc.group_refl, np.float64),
      group_abslen=_arr(sc.group_abslen, np.float64), group_record=_arr(sc.group_record, np.int32),
      group_grating_type=_arr(sc.group_grating_type, np.int32),
      group_grating_lpm=_arr(sc.group_grating_lpm, np.float64),
      group_grating_dir=_arr(sc.group_grating_dir, np.float64),
      group_grating_order=_arr(sc.group_grating_order, np.int32),
      seq_mask=_arr(sc.seq_mask, np.uint64))
  if getattr(sc, 'tri_normals', None) is not None:
    keep_tri = _arr(sc.tri_normals, np.float64).reshape(-1, 9)
    if len(keep_tri) != len(sc.prim_type):
      raise ValueError('tri_normals needs one row of 9 values per primitive')
  d = SceneDesc()
  d.n_prims = len(k['prim_type'])
  d.n_conds = len(k['cond_prim'])
  d.n_groups = len(k['group_type'])
  for name, typ in SceneDesc._fields_:
    if name in k:
      setattr(d, name, _p(k[name], typ))
  d.seq_enabled = int(sc.seq_enabled)
  d.seq_len = len(k['seq_mask'])
  d.ignore_mask = int(sc.ignore_mask)
  if getattr(sc, 'tri_normals', None) is not None:
    k['tri_normals'] = keep_tri
    d.tri_normals = _p(keep_tri, _pd)
  if getattr(sc, 'tri_edges', None) is not None:
    k['tri_edges'] = _arr(sc.tri_edges, np.int32)
    if len(k['tri_edges']) != len(sc.prim_type):
      raise ValueError('tri_edges needs one entry per primitive')
    d.tri_edges = k['tri_edges'].ctypes.data_as(_pi)
  return _Keep(d, k)


def source_desc(src):
  t = src.tables
  k = dict(phi_edges=_arr(t.phi_edges, np.float64), phi_cdf=_arr(t.phi_cdf, np.float64),
           t_edges=_arr(t.t_edges, np.float64), t_cdf=_arr(t.t_cdf, np.float64))
  d = SourceDesc()
  d.xform = (C.c_double * 12)(*np.asarray(src.xform, dtype=np.float64).reshape(12))
  d.focal_length = float(src.focal_length)
  d.wavelength = float(src.wavelength)
  d.power = float(src.power)
  d.n_phi_knots = len(k['phi_edges'])
  d.n_t_knots = len(k['t_edges'])
  d.n_t_rows = k['t_cdf'].shape[0] if k['t_cdf'].ndim == 2 else 1
  for name in k:
    setattr(d, name, _p(k[name], _pd))
  return _Keep(d, k)


def limits_desc(lim):
  return LimitsDesc(float(lim.max_ray_length), int(lim.max_intersections), float(lim.dist_tol),
                    float(lim.power_tol))


def detector_desc(det):
  if det is None:
    return None
  d = DetectorDesc()
  d.group = int(det['group'])
  d.origin = (C.c_double * 3)(*det['origin'])
  d.ex = (C.c_double * 3)(*det['ex'])
  d.ey = (C.c_double * 3)(*det['ey'])
  d.x_lo, d.x_hi, d.y_lo, d.y_hi = (float(det[k]) for k in ('x_lo', 'x_hi', 'y_lo', 'y_hi'))
  d.nx, d.ny = int(det['nx']), int(det['ny'])
  return d


_surface_keep = None


def set_surface_samplers(samplers, explicit_ray_seed=0):
  """register the stochastic-surface tables (scene.surface_samplers) for the
  following trace calls; None / [] clears"""
  global _surface_keep
  samplers = list(samplers or [])
  arr = (SurfaceSamplerDesc * max(1, len(samplers)))()
  keep = [arr]
  for d, s in zip(arr, samplers):
    phi_edges, phi_cdf = _arr(s.phi_edges, np.float64), _arr(s.phi_cdf, np.float64)
    t_edges, t_cdf = _arr(s.t_edges, np.float64), _arr(s.t_cdf, np.float64)
    keep += [phi_edges, phi_cdf, t_edges, t_cdf]
    d.group, d.kind, d.family_axis, d.n_family = int(s.group), int(s.kind), int(s.axis), int(s.n_family)
    d.family_lo, d.family_hi = float(s.lo), float(s.hi)
    d.n_phi_knots, d.n_t_knots, d.n_t_rows = len(phi_edges), len(t_edges), int(t_cdf.shape[-2])
    d.phi_edges, d.phi_cdf, d.t_edges, d.t_cdf = (_p(a, _pd) for a in (phi_edges, phi_cdf, t_edges, t_cdf))
    d.mu = float(getattr(s, 'mu', 0.0))
    d.n_atoms = int(getattr(s, 'n_atoms', 0) or 0)
    if d.n_atoms:
      am, at, ap = (_arr(getattr(s, n), np.float64) for n in ('atom_mass', 'atom_theta', 'atom_phi'))
      keep += [am, at, ap]
      d.atom_mass, d.atom_theta, d.atom_phi = _p(am, _pd), _p(at, _pd), _p(ap, _pd)
  _surface_keep = keep
  lib().odw_oracle_set_surface_samplers(arr, C.c_int32(len(samplers)), C.c_uint64(int(explicit_ray_seed)))


def scatter(group, ray, seed, ordinal, din, ideal, normal, mu=0.0):
  a, b, c = ((C.c_double * 3)(*v) for v in (din, ideal, normal))
  out = (C.c_double * 3)()
  lib().odw_oracle_scatter(C.c_int(group), C.c_uint64(ray), C.c_uint64(seed), C.c_uint32(ordinal), a, b, c, C.c_double(mu), out)
  return np.array(out[:])


def philox(ctr, key):
  c = (C.c_uint32 * 4)(*ctr)
  k = (C.c_uint32 * 2)(*key)
  o = (C.c_uint32 * 4)()
  lib().odw_oracle_philox(c, k, o)
  return list(o)


def interp(x, xp, fp):
  xp = _arr(xp, np.float64)
  fp = _arr(fp, np.float64)
  return np.array([lib().odw_oracle_interp(float(v), _p(xp, _pd), _p(fp, _pd), len(xp))
                   for v in np.atleast_1d(x)])


def sample(src, first, n, seed):
  s = source_desc(src)
  t = np.empty(n)
  phi = np.empty(n)
  lib().odw_oracle_sample(C.byref(s.desc), C.c_uint64(first), C.c_uint64(n), C.c_uint64(seed),
                          _p(t, _pd), _p(phi, _pd))
  return t, phi


def sample_uniforms(src, u_phi, u_t):
  s = source_desc(src)
  u_phi = _arr(u_phi, np.float64)
  u_t = _arr(u_t, np.float64)
  t = np.empty(len(u_phi))
  phi = np.empty(len(u_phi))
  lib().odw_oracle_sample_uniforms(C.byref(s.desc), C.c_uint64(len(u_phi)), _p(u_phi, _pd),
                                   _p(u_t, _pd), _p(t, _pd), _p(phi, _pd))
  return t, phi


def make_rays(src, first, n, seed):
  s = source_desc(src)
  o = np.empty((n, 3))
  d = np.empty((n, 3))
  lib().odw_oracle_make_rays(C.byref(s.desc), C.c_uint64(first), C.c_uint64(n), C.c_uint64(seed),
                             _p(o, _pd), _p(d, _pd))
  return o, d


def _result(hits, nh, hist, cnt, det):
  out = dict(hits=hits[:nh.value], counters={k: int(v) for k, v in zip(CNT_NAMES, cnt)})
  if hist is not None:
    out['hist'] = hist.reshape(det['nx'], det['ny'])
  return out


def trace(sc, src, lim, first, n, seed, det=None, flags=TRACE_RECORD_HITS | TRACE_HISTOGRAM,
          hit_capacity=None, nthreads=1):
  s, q, l, d = scene_desc(sc), source_desc(src), limits_desc(lim), detector_desc(det)
  set_surface_samplers(getattr(sc, 'surface_samplers', None))
  cap = int(hit_capacity if hit_capacity is not None else max(16, 2 * n))
  hits = np.zeros(cap, dtype=HIT_DTYPE)
  nh = C.c_uint64(0)
  hist = np.zeros(det['nx'] * det['ny'], dtype=np.uint64) if det is not None else None
  cnt = np.zeros(len(CNT_NAMES), dtype=np.uint64)
  rc = lib().odw_oracle_trace(
      C.byref(s.desc), C.byref(q.desc), C.byref(l), C.byref(d) if d is not None else None,
      C.c_uint64(first), C.c_uint64(n), C.c_uint64(seed), C.c_uint32(flags),
      hits.ctypes.data_as(C.c_void_p), C.c_uint64(cap), C.byref(nh),
      _p(hist, _pu) if hist is not None else None, _p(cnt, _pu), C.c_int(nthreads))
  if rc != 0:
    raise RuntimeError(f'odw_oracle_trace failed: {rc}')
  return _result(hits, nh, hist, cnt, det)


def trace_rays(sc, lim, origins, dirs, powers=None, wavelength=500.0, first=0, det=None,
               flags=TRACE_RECORD_HITS | TRACE_HISTOGRAM, nthreads=1, surface_seed=0):
  s, l, d = scene_desc(sc), limits_desc(lim), detector_desc(det)
  set_surface_samplers(getattr(sc, 'surface_samplers', None), surface_seed)
  origins = _arr(origins, np.float64).reshape(-1, 3)
  dirs = _arr(dirs, np.float64).reshape(-1, 3)
  n = len(origins)
  pw = _arr(powers, np.float64) if powers is not None else None
  cap = max(16, n * max(1, int(lim.max_intersections)))
  hits = np.zeros(cap, dtype=HIT_DTYPE)
  nh = C.c_uint64(0)
  hist = np.zeros(det['nx'] * det['ny'], dtype=np.uint64) if det is not None else None
  cnt = np.zeros(len(CNT_NAMES), dtype=np.uint64)
  rc = lib().odw_oracle_trace_rays(
      C.byref(s.desc), C.byref(l), C.byref(d) if d is not None else None, C.c_double(wavelength),
      C.c_uint64(first), C.c_uint64(n), _p(origins, _pd), _p(dirs, _pd),
      _p(pw, _pd) if pw is not None else None, C.c_uint32(flags),
      hits.ctypes.data_as(C.c_void_p), C.c_uint64(cap), C.byref(nh),
      _p(hist, _pu) if hist is not None else None, _p(cnt, _pu), C.c_int(nthreads))
  if rc != 0:
    raise RuntimeError(f'odw_oracle_trace_rays failed: {rc}')
  return _result(hits, nh, hist, cnt, det)


def trace_segments(sc, lim, src=None, first=0, n=0, seed=0, origins=None, dirs=None, powers=None,
                   wavelength=500.0, surface_seed=0):
  """segments of every ray sorted by (ray, ordinal): rays first..first+n-1 of the
  sampler `src`, or explicit initial conditions"""
  s, l = scene_desc(sc), limits_desc(lim)
  q = source_desc(src) if src is not None else None
  set_surface_samplers(getattr(sc, 'surface_samplers', None), surface_seed)
  pw = None
  if origins is not None:
    origins = _arr(origins, np.float64).reshape(-1, 3)
    dirs = _arr(dirs, np.float64).reshape(-1, 3)
    n = len(origins)
    pw = _arr(powers, np.float64) if powers is not None else None
  cap = max(16, int(n) * max(1, int(lim.max_intersections)))
  segs = np.zeros(cap, dtype=SEGMENT_DTYPE)
  ns = C.c_uint64(0)
  cnt = np.zeros(len(CNT_NAMES), dtype=np.uint64)
  rc = lib().odw_oracle_trace_segments(
      C.byref(s.desc), C.byref(q.desc) if q is not None else None, C.byref(l), C.c_double(wavelength),
      C.c_uint64(first), C.c_uint64(n), C.c_uint64(seed),
      _p(origins, _pd) if origins is not None else None, _p(dirs, _pd) if origins is not None else None,
      _p(pw, _pd) if pw is not None else None,
      segs.ctypes.data_as(C.c_void_p), C.c_uint64(cap), C.byref(ns), _p(cnt, _pu))
  if rc != 0:
    raise RuntimeError(f'odw_oracle_trace_segments failed: {rc}')
  return dict(segments=segs[:ns.value], counters={k: int(v) for k, v in zip(CNT_NAMES, cnt)})


def nearest(sc, lim, start, direction, medium=-1, seq_idx=0):
  s, l = scene_desc(sc), limits_desc(lim)
  st = (C.c_double * 3)(*start)
  di = (C.c_double * 3)(*direction)
  prim, face, group = C.c_int(), C.c_int(), C.c_int()
  dist = C.c_double()
  pt = (C.c_double * 3)()
  nm = (C.c_double * 3)()
  found = lib().odw_oracle_nearest(C.byref(s.desc), C.byref(l), st, di, C.c_int(medium),
                                   C.c_int(seq_idx), C.byref(prim), C.byref(face), C.byref(group),
                                   C.byref(dist), pt, nm)
  if not found:
    return None
  return dict(prim=prim.value, face=face.value, group=group.value, dist=dist.value,
              point=np.array(pt[:]), normal=np.array(nm[:]))


def threads():
  return int(lib().odw_oracle_threads())


def set_strict(on):
  """reference-strict findNearestIntersection (odw_oracle.c, nearest_strict): the reference's own
  candidate order, maxRayLength shrink and selection, no convex-solid skip; returns the old mode"""
  old = bool(lib().odw_oracle_get_strict())
  lib().odw_oracle_set_strict(C.c_int(1 if on else 0))
  return old


class strict:
  """with capi.strict(): ... -- the oracle in reference-strict mode inside the block"""

  def __init__(self, on=True):
    self.on = on

  def __enter__(self):
    self.old = set_strict(self.on)
    return self

  def __exit__(self, *exc):
    set_strict(self.old)
    return False


def surface_source_desc(src):
  """odw_surface_source_desc for a freecad_elements.surface_source.BakedSurfaceSource"""
  keep = dict(prim_type=_arr(src.prim_type, np.int32), prim_flags=_arr(src.prim_flags, np.int32),
              prim_xform=_arr(src.prim_xform, np.float64), prim_params=_arr(src.prim_params, np.float64),
              prim_cond_off=_arr(src.prim_cond_off, np.int32),
              cond_prim=_arr(src.cond_prim if len(src.cond_prim) else [0], np.int32),
              cond_inside=_arr(src.cond_inside if len(src.cond_inside) else [0], np.int32),
              face_prim=_arr(src.face_prim, np.int32), face_id=_arr(src.face_id, np.int32),
              face_area=_arr(src.face_area, np.float64), t_edges=_arr(src.t_edges, np.float64),
              t_cdf=_arr(src.t_cdf, np.float64))
  if getattr(src, 'tri_normals', None) is not None:
    keep['tri_normals'] = _arr(src.tri_normals, np.float64).reshape(-1, 9)
    if len(keep['tri_normals']) != len(keep['prim_type']):
      raise ValueError('tri_normals needs one row of 9 values per primitive')
  d = SurfaceSourceDesc()
  d.wavelength, d.power, d.dist_tol = float(src.wavelength), float(src.power), float(src.dist_tol)
  d.n_prims, d.n_conds, d.n_faces = len(keep['prim_type']), len(src.cond_prim), len(keep['face_prim'])
  d.n_t_knots = len(keep['t_edges'])
  for name, typ in SurfaceSourceDesc._fields_:
    if name in keep:
      setattr(d, name, keep[name].ctypes.data_as(typ))
  return _Keep(d, keep)


def surface_rays(src, first, n, seed):
  """initial conditions of a surface source (odw_oracle_surface_rays)"""
  s = surface_source_desc(src)
  o = np.empty((n, 3))
  d = np.empty((n, 3))
  rc = lib().odw_oracle_surface_rays(C.byref(s.desc), C.c_uint64(first), C.c_uint64(n), C.c_uint64(seed),
                                     _p(o, _pd), _p(d, _pd))
  if rc != 0:
    raise RuntimeError(f'odw_oracle_surface_rays failed: {rc}')
  return o, d


def trace_surface(sc, src, lim, first, n, seed, **kw):
  """surface source + trace: rays `first`.. of stream `seed`, surface draws keyed by the same seed"""
  o, d = surface_rays(src, first, n, seed)
  return trace_rays(sc, lim, o, d, wavelength=src.wavelength, first=first, surface_seed=seed, **kw)
