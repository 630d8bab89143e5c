"""The HOST code of libodw_trace under AddressSanitizer + UndefinedBehaviorSanitizer: the entry points that never touch a
GPU -- scene validation and host tables, bounding boxes, the choice and construction of grid / binary tree / eight-wide
tree (odw_build_check: the same builders a context with a device runs, their tables in host memory), the plane screens
of detectPlaneNormal's search, the header of a scene-compiled kernel, and every entry point's answer to a null context.
The device code is compiled as usual (no GPU sanitizers on this pool).  Runs in a child process because the sanitizer
runtime must be the first library loaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.environ['ODW_ROOT']); sys.path.insert(0, os.path.join(os.environ['ODW_ROOT'], 'tests'))
import numpy as np
from freecad.optics_design_workbench_amd import _native, scenes
lib = C.CDLL(os.environ['ODW_ASAN_LIB'])
lib.odw_last_error.restype = C.c_char_p
lib.odw_last_error.argtypes = [C.c_void_p]
assert lib.odw_abi_version() == _native.ABI_VERSION
for name in _native.SYMBOLS:
  getattr(lib, name)
from conftest import project
import random_scenes

# -- builders: every structure, on reference scenes and on random ones ------------------------------------------------
want = dict(lensesAndMirrors='flat', GettingStarted='flat', hugeArray='grid', playground='flat', grating='flat')
want['imported-stepfile-as-surface-source'] = 'wide-bvh'
for name, structure in want.items():
  pr = project(name)
  got = _native.build_check(pr.scene, pr.limits, library=lib)
  assert got['structure'] == structure and got['primitives'] == len(pr.scene.prim_type), (name, got)
huge = _native.build_check(project('hugeArray').scene, project('hugeArray').limits, library=lib)
assert huge['grid_cells'] == 1500 and huge['grid_items'] == 1500 and 0 < huge['grid_lds_bytes'] <= 160 * 1024
seen = set()
rs = np.random.RandomState(5)
for k in range(24):
  sc, lim, _ = random_scenes.scene(rs, rich=(k % 3 == 0), crowded=(k % 2 == 1), paraboloids=(k % 4 == 3))
  got = _native.build_check(sc, lim, library=lib)
  seen.add(got['structure'])
  assert got['primitives'] == len(sc.prim_type)
assert {'flat', 'wide-bvh'} <= seen and ('grid' in seen or 'bvh' in seen), seen
# a strictly convex tessellated lens beside an analytic screen: the eight-wide tree with a normal cone per slot
# (WideBvh::cone_word / node_span), coarse and fine, at the tolerance that arms the cones and at one that does not
from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import Document, bake
for segments, tol in ((8, '1e-2'), (48, '1e-6'), (96, '1e-4')):
  doc = Document()
  ball = make.makeTessellated(doc, make.makeSphere(doc, 'S', 5, base=(0, 0, 30)), segments)
  make.makeLens(doc, [ball], RefractiveIndex=1.5)
  make.makeAbsorber(doc, [make.makeBox(doc, 'A', 100, 100, 1, base=(-50, -50, 60))])
  make.makeSimulationSettings(doc, DistanceTolerance=tol)
  src = make.makePointSource(doc, PowerDensity='1')
  sc, lim = bake.bakeScene(doc, src), bake.bakeLimits(doc, src)
  from freecad.optics_design_workbench_amd.scene import geometry
  assert (sc.prim_flags[sc.prim_type == geometry.TRIANGLE] & 8).all() and (sc.prim_type == geometry.TRIANGLE).sum() >= 48
  got = _native.build_check(sc, lim, library=lib)
  assert got['structure'] == 'wide-bvh' and got['primitives'] == len(sc.prim_type), got
# the flat loop's limit moved: the same crowded scene on the other analytic structures
os.environ['ODW_BVH_THRESHOLD'] = '4'
sc, lim, _ = random_scenes.scene(np.random.RandomState(7), crowded=True)
assert _native.build_check(sc, lim, library=lib)['structure'] in ('grid', 'bvh')
del os.environ['ODW_BVH_THRESHOLD']

# -- descriptors the validation must refuse (return codes, no crash) -----------------------------------------------------
import copy
pr = project('lensesAndMirrors')
def refused(mutate, limits=None):
  sc = copy.copy(pr.scene)
  mutate(sc)
  try:
    _native.build_check(sc, limits or pr.limits, library=lib)
  except _native.NativeError:
    return True
  return False
def set_attr(name, value):
  return lambda sc: setattr(sc, name, value)
bad_type = np.array(pr.scene.group_type).copy(); bad_type[0] = 99
assert refused(set_attr('group_type', bad_type))
bad_group = np.array(pr.scene.prim_group).copy(); bad_group[0] = 77
assert refused(set_attr('prim_group', bad_group))
bad_prim = np.array(pr.scene.prim_type).copy(); bad_prim[0] = 42
assert refused(set_attr('prim_type', bad_prim))
bad_cond = np.array(pr.scene.cond_prim).copy()
if len(bad_cond):
  bad_cond[0] = 10_000
  assert refused(set_attr('cond_prim', bad_cond))
import dataclasses
assert refused(lambda sc: None, dataclasses.replace(pr.limits, dist_tol=0.0))
assert refused(lambda sc: None, dataclasses.replace(pr.limits, max_ray_length=-1.0))
assert lib.odw_build_check(None, None, None, None) == 1

# -- the plane screens ---------------------------------------------------------------------------------------------------
pd = C.POINTER(C.c_double)
rng = np.random.default_rng(3)
cloud = np.ascontiguousarray(rng.normal(size=(300, 3)) * np.array([1.0, 0.5, 1e-9]))
phis, thetas = np.linspace(0, np.pi, 30), np.linspace(-np.pi / 2, np.pi / 2, 30)
ext = np.empty(900)
assert lib.odw_plane_screen(cloud.ctypes.data_as(pd), C.c_uint64(300), phis.ctypes.data_as(pd), C.c_int32(30), thetas.ctypes.data_as(pd),
                            C.c_int32(30), ext.ctypes.data_as(pd)) == 0
assert np.isfinite(ext).all() and 0 < ext.min() < 1.0
assert lib.odw_plane_screen(cloud.ctypes.data_as(pd), C.c_uint64(0), phis.ctypes.data_as(pd), C.c_int32(30), thetas.ctypes.data_as(pd),
                            C.c_int32(30), ext.ctypes.data_as(pd)) == 1
nan_cloud = cloud.copy(); nan_cloud[17, 1] = np.nan
assert lib.odw_plane_screen(nan_cloud.ctypes.data_as(pd), C.c_uint64(300), phis.ctypes.data_as(pd), C.c_int32(30), thetas.ctypes.data_as(pd),
                            C.c_int32(30), ext.ctypes.data_as(pd)) == 0 and np.isnan(ext).all()
clouds = [np.ascontiguousarray(rng.normal(size=(n, 3))) for n in (300, 7, 1, 250, 300, 299, 300, 120, 300, 300)]
ptrs = (pd * len(clouds))(*[c.ctypes.data_as(pd) for c in clouds])
counts = np.array([len(c) for c in clouds], dtype=np.uint64)
P, T = np.tile(phis, (len(clouds), 1)), np.tile(thetas, (len(clouds), 1))
out = np.empty((len(clouds), 900))
assert lib.odw_plane_screen_batch(ptrs, counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(len(clouds)), P.ctypes.data_as(pd), C.c_int32(30),
                                  T.ctypes.data_as(pd), C.c_int32(30), out.ctypes.data_as(pd)) == 0
for k, c in enumerate(clouds):
  one = np.empty(900)
  lib.odw_plane_screen(c.ctypes.data_as(pd), C.c_uint64(len(c)), phis.ctypes.data_as(pd), C.c_int32(30), thetas.ctypes.data_as(pd), C.c_int32(30),
                       one.ctypes.data_as(pd))
  assert np.array_equal(one, out[k])
assert lib.odw_plane_screen_batch(None, None, C.c_int32(2), None, C.c_int32(30), None, C.c_int32(30), None) == 1

# -- the header of a scene-compiled kernel (hiprtc is another library: only our side of it is instrumented) ----------------
d, keep = _native.scene_desc(pr.scene)
lim = _native.LimitsDesc(float(pr.limits.max_ray_length), int(pr.limits.max_intersections), float(pr.limits.dist_tol), float(pr.limits.power_tol))
buf = C.create_string_buffer(1 << 20)
size = C.c_uint64(0)
f = lib.odw_compile_check
f.argtypes = [C.POINTER(_native.SceneDesc), C.POINTER(_native.LimitsDesc), C.c_int32, C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
assert f(C.byref(d), C.byref(lim), 1, None, buf, len(buf), C.byref(size)) == 0 and size.value > 10000 and b'ODW_SPEC' in buf.value
small = C.create_string_buffer(64)                       # a header buffer that is too small is cut, not overrun
assert f(C.byref(d), C.byref(lim), 1, None, small, len(small), C.byref(size)) == 0 and len(small.value) == 63
huge_pr = project('hugeArray')
dh, keep_h = _native.scene_desc(huge_pr.scene)
assert f(C.byref(dh), C.byref(lim), 1, None, buf, len(buf), C.byref(size)) == 5         # outside the flat kernels' domain

# -- every entry point that takes a context answers a null one with a code ------------------------------------------------
ctx = C.c_void_p(0)
z64, zi = C.c_uint64(0), C.c_int32(0)
assert lib.odw_create(C.c_int(-1), C.byref(ctx)) in (1, 2) and not ctx.value
for name, args in (('odw_upload_scene', (None, None)), ('odw_set_limits', (None, None)), ('odw_set_detector', (None, None)),
                   ('odw_trace', (None, z64, z64, z64, C.c_uint32(0))), ('odw_trace_batch', (None, z64, z64, z64, C.c_uint32(0), z64)),
                   ('odw_batch_reserve', (None, zi, z64, z64)), ('odw_batch_hits_begin', (None, zi, z64)),
                   ('odw_batch_hits_sampled', (None, zi, None, None, None, None, z64, None)),
                   ('odw_batch_hits_measure', (None, None, None, None, zi, None, zi, None, zi, z64)),
                   ('odw_batch_hits_measured', (None, zi, None, None, None, None, None, None, z64, None)),
                   ('odw_batch_hits_select', (None, zi, None, None, None)), ('odw_hits_select', (None, zi, None, None)),
                   ('odw_reserve_hits', (None, z64)), ('odw_sync', (None,)), ('odw_fetch_counters', (None, None, zi)),
                   ('odw_compile_scene', (None, zi)), ('odw_upload_scene_batch', (None, None, zi)), ('odw_archive_reset', (None,))):
  rc = getattr(lib, name)(*args)
  assert rc != 0, name
lib.odw_destroy(None)
print('SANITIZED-OK')
'''


def test_host_code_of_the_library_under_asan_ubsan():
  from freecad.optics_design_workbench_amd import _native
  so, runtime = _native.build_sanitized()
  env = dict(os.environ, ODW_ROOT=ROOT, ODW_ASAN_LIB=so, LD_PRELOAD=runtime,
             ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
  res = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=900)
  assert res.returncode == 0 and 'SANITIZED-OK' in res.stdout, (res.stdout[-2000:], res.stderr[-6000:])
  assert 'runtime error' not in res.stderr and 'AddressSanitizer' not in res.stderr, res.stderr[-6000:]
