"""ctypes binding of the HIP library (csrc/libodw_trace.so, include/odw_trace.h).

There is no CPU fallback: if the library is missing or no GPU is present the
calls raise.  `build()` compiles the library in-tree with hipcc for gfx950
(cross-compiles without a GPU).
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

# HIP spreads a process' streams over this many hardware queues (default 4); kernels of streams that share one wait for each
# other.  A parameter sweep keeps five contexts' chains in flight beside each other: small kernels of one must not queue
# behind the 7 ms launch of another.  Read by the HIP runtime when it initialises, so: set before the first HIP call.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(CSRC, 'libodw_trace.so')
_SOURCES = ['odw_capi.hip', 'odw_kernels.hip', 'odw_grid.hip', 'odw_mesh.hip', 'odw_posthoc.hip', 'odw_posthoc_batch.hip', 'odw_spec.hip', 'odw_device.h']
_HEADER = os.path.normpath(os.path.join(_HERE, '..', '..', 'include', 'odw_trace.h'))

ABI_VERSION = 10
CNT_NAMES = ['traced_rays', 'recorded_hits', 'segments', 'escaped', 'died', 'capped',
             'hist_overflow', 'hits_dropped', 'grating_in_medium']
TRACE_RECORD_HITS, TRACE_HISTOGRAM, TRACE_RECORD_SEGMENTS = 1, 2, 4
FLAG_FLIP_NORMAL, FLAG_CONVEX = 1, 2      # ODW_FLAG_* of prim_flags (include/odw_trace.h)
COMPILE_OFF, COMPILE_STRUCTURE, COMPILE_AUTO = 0, 1, 2
COMPILE_MODES = {None: 0, False: 0, 'off': 0, 0: 0, 'structure': 1, 1: 1, True: 1, 'auto': 2, 2: 2}
ERRORS = {1: 'invalid argument', 2: 'device error', 3: 'no scene', 4: 'capacity', 5: 'unsupported'}
BUSY = 6             # ODW_BUSY: not an error (polling calls)

HIT_DTYPE = np.dtype([('point', '<f8', 3), ('direction', '<f8', 3), ('power', '<f8'), ('tag', '<u8')])
SEGMENT_DTYPE = np.dtype([('p1', '<f8', 3), ('p2', '<f8', 3), ('power', '<f8'), ('tag', '<u8')])

# every symbol include/odw_trace.h declares
SYMBOLS = ['odw_abi_version', 'odw_create', 'odw_destroy', 'odw_last_error', 'odw_upload_scene',
           'odw_upload_source', 'odw_upload_surface_source', 'odw_generate_rays', 'odw_upload_surface_samplers', 'odw_set_surface_seed', 'odw_set_wavelength', 'odw_set_limits', 'odw_set_detector', 'odw_reserve_hits', 'odw_reserve_segments', 'odw_trace',
           'odw_trace_rays', 'odw_sync', 'odw_reset_results', 'odw_reset_hits', 'odw_fetch_counters', 'odw_hit_count',
           'odw_fetch_hits', 'odw_fetch_histogram', 'odw_segment_count', 'odw_fetch_segments', 'odw_reset_segments', 'odw_sample', 'odw_device_histogram',
           'odw_device_counters', 'odw_device_results', 'odw_stream', 'odw_timing_enable', 'odw_timing_read',
           'odw_swap_hit_lists', 'odw_fetch_swapped_hits', 'odw_release_swapped_hits', 'odw_mem_info', 'odw_host_alloc', 'odw_host_free', 'odw_load_hits', 'odw_hits_select', 'odw_hits_gather', 'odw_hits_project', 'odw_hits_range', 'odw_hits_bin', 'odw_hits_moments', 'odw_plane_screen',
           'odw_compile_scene', 'odw_compiled_info', 'odw_compile_check', 'odw_build_check', 'odw_hits_columns',
           'odw_upload_scene_batch', 'odw_trace_batch', 'odw_batch_select', 'odw_batch_rows',
           'odw_plane_screen_batch', 'odw_archive_append', 'odw_archive_select', 'odw_archive_reset', 'odw_batch_hits_select', 'odw_batch_hits_sample', 'odw_batch_hits_project', 'odw_batch_hits_bin',
           'odw_batch_reserve', 'odw_batch_hits_begin', 'odw_batch_hits_sampled', 'odw_batch_hits_measure', 'odw_batch_hits_measured']

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


class SurfaceSamplerDesc(C.Structure):
  _fields_ = [('group', C.c_int32), ('kind', C.c_int32), ('family_axis', C.c_int32), ('n_family', C.c_int32),
              ('family_lo', C.c_double), ('family_hi', C.c_double), ('n_phi_knots', C.c_int32),
              ('phi_edges', _pd), ('phi_cdf', _pd), ('n_t_knots', C.c_int32), ('n_t_rows', C.c_int32),
              ('t_edges', _pd), ('t_cdf', _pd), ('mu', C.c_double), ('n_atoms', C.c_int32),
              ('atom_mass', _pd), ('atom_theta', _pd), ('atom_phi', _pd)]


class SurfaceSourceDesc(C.Structure):
  _fields_ = [('wavelength', C.c_double), ('power', C.c_double), ('dist_tol', C.c_double),
              ('n_prims', C.c_int32), ('prim_type', _pi), ('prim_flags', _pi), ('prim_xform', _pd),
              ('prim_params', _pd), ('prim_cond_off', _pi), ('n_conds', C.c_int32), ('cond_prim', _pi),
              ('cond_inside', _pi), ('n_faces', C.c_int32), ('face_prim', _pi), ('face_id', _pi),
              ('face_area', _pd), ('n_t_knots', C.c_int32), ('t_edges', _pd), ('t_cdf', _pd), ('tri_normals', _pd)]


class LimitsDesc(C.Structure):
  _fields_ = [('max_ray_length', C.c_double), ('max_intersections', C.c_int32),
              ('dist_tol', C.c_double), ('power_tol', C.c_double)]


class DetectorDesc(C.Structure):
  _fields_ = [('group', C.c_int32), ('origin', C.c_double * 3), ('ex', C.c_double * 3),
              ('ey', C.c_double * 3), ('x_lo', C.c_double), ('x_hi', C.c_double),
              ('y_lo', C.c_double), ('y_hi', C.c_double), ('nx', C.c_int32), ('ny', C.c_int32)]


class NativeError(RuntimeError):
  pass


def hipcc():
  for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
    if cand and os.path.exists(cand):
      return cand
  raise NativeError('hipcc not found; the HIP library cannot be built')


def sources_hash(csrc=None, header=None):
  """sha256 over the sources a build of the library is made of (csrc/*.hip, csrc/*.h, include/odw_trace.h; names and
  contents, in name order): what ties a committed counter pass (profiles/pmc_current.json) to the code it was taken on"""
  import hashlib
  csrc = CSRC if csrc is None else csrc
  header = _HEADER if header is None else header
  h = hashlib.sha256()
  for name in sorted(_SOURCES):
    path = os.path.join(csrc, name)
    h.update(name.encode() + b'\0')
    with open(path, 'rb') as f:
      h.update(f.read())
    h.update(b'\0')
  h.update(b'odw_trace.h\0')
  with open(header, 'rb') as f:
    h.update(f.read())
  return h.hexdigest()


def needs_build():
  if not os.path.exists(LIB_PATH):
    return True
  t = os.path.getmtime(LIB_PATH)
  deps = [os.path.join(CSRC, s) for s in _SOURCES] + [_HEADER]
  return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
  """compile csrc/ for gfx950 into csrc/libodw_trace.so"""
  if not force and not needs_build():
    return LIB_PATH
  # -ffp-contract=on: products and sums are fused where one expression says so, never across statements
  # -- the generic kernels and the scene-compiled ones (odw_spec.hip, same flag) then round alike
  cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '-fPIC', '-shared',
         '-o', LIB_PATH + '.tmp', os.path.join(CSRC, 'odw_capi.hip')]
  if verbose:
    cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
  res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
  if res.returncode != 0:
    raise NativeError('hipcc failed:\n' + res.stdout + res.stderr)
  os.replace(LIB_PATH + '.tmp', LIB_PATH)
  if verbose:
    print(res.stderr)
  return LIB_PATH


ASAN_LIB_PATH = os.path.join(CSRC, 'libodw_trace_asan.so')


def build_sanitized(force=False):
  """The library with AddressSanitizer + UndefinedBehaviorSanitizer on its HOST code (the device code is compiled as
  usual: GPU sanitizers are not available on this pool), for the entry points that never touch a GPU -- scene validation,
  boxes, the grid / tree builders (odw_build_check), the plane screens, the header of a compiled scene -- in a child
  process with the sanitizer runtime preloaded (tests/test_native_sanitized.py).  -> (library path, runtime path)"""
  runtime = subprocess.run([hipcc(), '--print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True).stdout.strip()
  if not os.path.isabs(runtime) or not os.path.exists(runtime):
    raise NativeError('the sanitizer runtime of hipcc\'s clang was not found')
  deps = [os.path.join(CSRC, s) for s in _SOURCES] + [_HEADER]
  if force or not os.path.exists(ASAN_LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(ASAN_LIB_PATH) for d in deps):
    cmd = [hipcc(), '--offload-arch=gfx950', '-O1', '-g', '-std=c++17', '-ffp-contract=on', '-fPIC', '-shared',
           '-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-omit-frame-pointer', '-shared-libasan',
           '-o', ASAN_LIB_PATH + '.tmp', os.path.join(CSRC, 'odw_capi.hip')]
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
      raise NativeError('hipcc (sanitized build) failed:\n' + res.stdout + res.stderr)
    os.replace(ASAN_LIB_PATH + '.tmp', ASAN_LIB_PATH)
  return ASAN_LIB_PATH, runtime


_lib = None


def lib():
  """the loaded library; raises if it is absent (no fallback)"""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise NativeError(f'{LIB_PATH} is missing: run __graft_entry__.build() '
                        f'(or freecad.optics_design_workbench_amd._native.build()) first')
    # ODW_TRACE_LIB: another build of the same library (kernel experiments, scripts/try_variants.sh)
    l = C.CDLL(os.environ.get('ODW_TRACE_LIB') or LIB_PATH)
    l.odw_last_error.restype = C.c_char_p
    l.odw_last_error.argtypes = [C.c_void_p]
    l.odw_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    l.odw_destroy.argtypes = [C.c_void_p]
    l.odw_destroy.restype = None
    for name in SYMBOLS:
      getattr(l, name)
    if l.odw_abi_version() != ABI_VERSION:
      raise NativeError('libodw_trace.so ABI version mismatch; rebuild')
    _lib = l
  return _lib


def check(ctx, rc, what):
  if rc != 0:
    msg = lib().odw_last_error(ctx)
    raise NativeError(f'{what}: {ERRORS.get(rc, rc)}: {msg.decode() if msg else ""}')


def _arr(a, dtype):
  return np.ascontiguousarray(a, dtype=dtype)


def scene_desc(sc):
  keep = dict(
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
  d.n_prims, d.n_conds, d.n_groups = len(keep['prim_type']), len(keep['cond_prim']), len(keep['group_type'])
  for name, typ in SceneDesc._fields_:
    if name in keep:
      setattr(d, name, keep[name].ctypes.data_as(typ))
  d.seq_enabled, d.seq_len, d.ignore_mask = int(sc.seq_enabled), len(keep['seq_mask']), int(sc.ignore_mask)
  if getattr(sc, 'tri_normals', None) is not None:
    keep['tri_normals'] = keep_tri
    d.tri_normals = keep_tri.ctypes.data_as(_pd)
  if getattr(sc, 'tri_edges', None) is not None:
    keep['tri_edges'] = _arr(sc.tri_edges, np.int32)
    if len(keep['tri_edges']) != len(sc.prim_type):
      raise ValueError('tri_edges needs one entry per primitive')
    d.tri_edges = keep['tri_edges'].ctypes.data_as(_pi)
  return d, keep


def compile_check(scene, limits, mode='structure', arch=None):
  """Host only (no GPU): the header of constants the library writes for `scene` and the size of the
  code object hiprtc builds from it for `arch` (default gfx950).  Raises NativeError when the scene
  is outside the flat kernel's domain or the compilation fails."""
  d, keep = scene_desc(scene)
  lim = LimitsDesc(float(limits.max_ray_length), int(limits.max_intersections), float(limits.dist_tol),
                   float(limits.power_tol))
  buf = C.create_string_buffer(1 << 20)
  size = C.c_uint64(0)
  f = lib().odw_compile_check
  f.argtypes = [C.POINTER(SceneDesc), C.POINTER(LimitsDesc), C.c_int32, C.c_char_p, C.c_char_p, C.c_uint64,
                C.POINTER(C.c_uint64)]
  rc = f(C.byref(d), C.byref(lim), COMPILE_MODES[mode], arch.encode() if arch else None, buf, len(buf), C.byref(size))
  check(None, rc, 'odw_compile_check')
  return buf.value.decode(), int(size.value)


STRUCTURES = {0: 'flat', 1: 'grid', 2: 'bvh', 3: 'wide-bvh'}


def build_check(scene, limits, library=None):
  """Host only (no GPU): what the library will trace `scene` with -- 'flat' loop, rectilinear 'grid', binary 'bvh' or the
  eight-wide tree of the mesh kernel --, after building it in host memory (`odw_build_check`), and the sizes it reports"""
  d, keep = scene_desc(scene)
  lim = LimitsDesc(float(limits.max_ray_length), int(limits.max_intersections), float(limits.dist_tol),
                   float(limits.power_tol))
  structure = C.c_int32(-1)
  sizes = (C.c_uint64 * 6)()
  f = (library or lib()).odw_build_check
  f.argtypes = [C.POINTER(SceneDesc), C.POINTER(LimitsDesc), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
  rc = f(C.byref(d), C.byref(lim), C.byref(structure), sizes)
  if rc != 0:
    msg = (library or lib()).odw_last_error(None)
    raise NativeError(f'odw_build_check: {ERRORS.get(rc, rc)}: {msg.decode() if msg else ""}')
  names = ('primitives', 'nodes', 'grid_cells', 'grid_items', 'grid_lds_bytes', 'dead_primitives')
  return dict(structure=STRUCTURES[int(structure.value)], **{k: int(v) for k, v in zip(names, sizes)})


def source_desc(src):
  t = src.tables
  keep = dict(phi_edges=_arr(t.phi_edges, np.float64), phi_cdf=_arr(t.phi_cdf, np.float64),
              t_edges=_arr(t.t_edges, np.float64), t_cdf=_arr(t.t_cdf, np.float64).reshape(-1, len(t.t_edges)))
  d = SourceDesc()
  d.xform = (C.c_double * 12)(*np.asarray(src.xform, dtype=np.float64).reshape(12))
  d.focal_length, d.wavelength, d.power = float(src.focal_length), float(src.wavelength), float(src.power)
  d.n_phi_knots, d.n_t_knots, d.n_t_rows = len(keep['phi_edges']), len(keep['t_edges']), keep['t_cdf'].shape[0]
  for name in keep:
    setattr(d, name, keep[name].ctypes.data_as(_pd))
  return d, keep


def surface_sampler_descs(samplers):
  """array of odw_surface_sampler_desc for scene.surface_samplers"""
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
    for name, a in (('phi_edges', phi_edges), ('phi_cdf', phi_cdf), ('t_edges', t_edges), ('t_cdf', t_cdf)):
      setattr(d, name, a.ctypes.data_as(_pd))
    d.mu = float(getattr(s, 'mu', 0.0))
    d.n_atoms = int(getattr(s, 'n_atoms', 0) or 0)
    if d.n_atoms:
      for name in ('atom_mass', 'atom_theta', 'atom_phi'):
        a = _arr(getattr(s, name), np.float64)
        keep.append(a)
        setattr(d, name, a.ctypes.data_as(_pd))
  return arr, len(samplers), keep


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
  return d, keep
