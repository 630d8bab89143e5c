"""`DeviceHits._flattest_direction` (the plane search of detectPlaneNormal with the library's screen,
`odw_plane_screen`: host code, no GPU) against the host routine `jupyter_utils.hits._flattest_direction`, which
tests/test_hits_histogram.py pins to the reference's outputs: the same normal bit for bit, also where candidates tie
(a plane across a coordinate axis: (phi, theta) and (phi, -theta) have the same extent up to rounding)."""
import ctypes as C
import types

import numpy as np
import pytest

from freecad.optics_design_workbench_amd.jupyter_utils import hits as host_hits
from freecad.optics_design_workbench_amd.simulation.device_hits import DeviceHits


def _device_hits(lib):
  h = DeviceHits.__new__(DeviceHits)
  h._tr = types.SimpleNamespace(_lib=lib)
  return h


def _clouds(rng, count):
  for k in range(count):
    n = int(rng.integers(3, 300))
    nrm = rng.normal(size=3)
    nrm /= np.linalg.norm(nrm)
    pts = rng.normal(size=(n, 3)) * rng.uniform(0.01, 50)
    kind = k % 5
    if kind == 0:
      pts -= np.outer(pts @ nrm, nrm)                       # a tilted plane (up to rounding)
    elif kind == 1:
      pts -= np.outer(pts @ nrm, nrm) * (1 - 1e-6)          # a thin slab
    elif kind == 2:
      pts[:, 2] = 15.0                                      # the detector of the BASELINE scenes: z = const
    elif kind == 3:
      pts[:, 0] = -3.25                                     # x = const
    if kind not in (2, 3):
      pts += rng.normal(size=3) * rng.uniform(0, 100)
    yield kind, np.ascontiguousarray(pts)


def test_screened_plane_search_is_the_host_search(native_lib):
  h = _device_hits(native_lib)
  rng = np.random.default_rng(20261004)
  for kind, cloud in _clouds(rng, 150):
    got = h._flattest_direction(cloud, 1e-9)
    want = host_hits._flattest_direction(cloud, 1e-9)
    assert got.tobytes() == want.tobytes(), (kind, got, want)


def test_screen_is_the_extent_per_candidate(native_lib):
  rng = np.random.default_rng(7)
  cloud = np.ascontiguousarray(rng.normal(size=(200, 3)) * [3.0, 1.0, 0.01] + [5.0, -2.0, 40.0])
  phis, thetas = np.linspace(0, np.pi, 30), np.linspace(-np.pi / 2, np.pi / 2, 30)
  out = np.empty(900)
  pd = C.POINTER(C.c_double)
  assert native_lib.odw_plane_screen(cloud.ctypes.data_as(pd), C.c_uint64(len(cloud)), phis.ctypes.data_as(pd), C.c_int32(30),
                                     thetas.ctypes.data_as(pd), C.c_int32(30), out.ctypes.data_as(pd)) == 0
  _, _, normals = host_hits._sphere_dirs(phis, thetas)
  along = cloud @ normals.T
  want = along.max(axis=0) - along.min(axis=0)
  assert np.abs(out - want).max() < 1e-12 * np.abs(cloud).max()
  assert native_lib.odw_plane_screen(cloud.ctypes.data_as(pd), C.c_uint64(0), phis.ctypes.data_as(pd), C.c_int32(30),
                                     thetas.ctypes.data_as(pd), C.c_int32(30), out.ctypes.data_as(pd)) != 0


def test_a_cloud_with_a_nan_fails_like_the_host_search(native_lib):
  h = _device_hits(native_lib)
  cloud = np.ones((10, 3))
  cloud[3, 1] = np.nan
  with pytest.raises(ValueError):
    host_hits._flattest_direction(cloud, 1e-9)
  with pytest.raises(ValueError):
    h._flattest_direction(cloud, 1e-9)


def test_linspace_of_ten_is_numpy_linspace():
  from freecad.optics_design_workbench_amd.simulation.device_hits import _linspace10
  rng = np.random.default_rng(5)
  for _ in range(2000):
    a = np.float64(rng.normal() * 10.0 ** rng.integers(-12, 3))
    w = np.float64(abs(rng.normal()) * 10.0 ** rng.integers(-14, 1))
    assert _linspace10(a - w, a + w).tobytes() == np.linspace(a - w, a + w, 10).tobytes()
  assert _linspace10(np.float64(1.5), np.float64(1.5)).tobytes() == np.linspace(1.5, 1.5, 10).tobytes()


def test_searches_in_lockstep_are_the_single_searches(native_lib):
  """`flattestDirections` (the plane searches of a batch of segments, level by level, one library call per level:
  odw_plane_screen_batch) against `flattestDirection` cloud by cloud: the same normals bit for bit -- tilted planes,
  thin slabs, axis-aligned planes (ties between candidates), blobs; groups of one to nine clouds, clouds of one point"""
  from freecad.optics_design_workbench_amd.simulation.device_hits import flattestDirection, flattestDirections
  rng = np.random.default_rng(424242)
  pool = [c for _, c in _clouds(rng, 120)]
  pool.append(np.array([[1.0, 2.0, 3.0]]))                     # one point: every direction is flat
  pool.append(np.zeros((5, 3)))
  pos = 0
  while pos < len(pool):
    size = int(rng.integers(1, 10))
    group = pool[pos:pos + size]
    pos += size
    got = flattestDirections(native_lib, group, 1e-9)
    assert len(got) == len(group)
    for cloud, g in zip(group, got):
      want = flattestDirection(native_lib, cloud, 1e-9)
      assert g.tobytes() == want.tobytes(), (len(cloud), g, want)
  # a cloud with a nan fails the batch like it fails its own search
  bad = np.ones((10, 3))
  bad[3, 1] = np.nan
  with pytest.raises(ValueError):
    flattestDirections(native_lib, [pool[0], bad, pool[1]], 1e-9)
  assert flattestDirections(native_lib, [], 1e-9) == []
