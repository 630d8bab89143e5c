"""The C-ABI library loads and exports every symbol include/odw_trace.h
declares; ctypes mirrors have the header's layout.  No compute calls (no GPU)."""
import ctypes as C
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'odw_trace.h')


def declared_functions():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(odw_[a-z_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported(native_lib):
  names = declared_functions()
  assert len(names) >= 20
  for n in names:
    assert hasattr(native_lib, n), n


def test_binding_lists_match_header():
  from freecad.optics_design_workbench_amd import _native
  assert sorted(_native.SYMBOLS) == declared_functions()


def test_abi_version(native_lib):
  from freecad.optics_design_workbench_amd import _native
  assert native_lib.odw_abi_version() == _native.ABI_VERSION


def test_struct_layouts():
  from freecad.optics_design_workbench_amd import _native
  assert _native.HIT_DTYPE.itemsize == 64
  assert C.sizeof(_native.LimitsDesc) == 32
  assert C.sizeof(_native.DetectorDesc) == 8 + 9 * 8 + 4 * 8 + 8
  assert C.sizeof(_native.SourceDesc) == 12 * 8 + 3 * 8 + 8 + 16 + 8 + 16
  from oracle import capi
  for a, b in ((capi.SceneDesc, _native.SceneDesc), (capi.SourceDesc, _native.SourceDesc),
               (capi.LimitsDesc, _native.LimitsDesc), (capi.DetectorDesc, _native.DetectorDesc),
               (capi.SurfaceSamplerDesc, _native.SurfaceSamplerDesc),
               (capi.SurfaceSourceDesc, _native.SurfaceSourceDesc)):
    assert C.sizeof(a) == C.sizeof(b)
    assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_]


def test_no_gpu_is_an_error_not_a_fallback(native_lib):
  """without a device odw_create fails loudly (this container has no GPU);
  with one it must succeed -- either way nothing is computed on the CPU"""
  ctx = C.c_void_p()
  rc = native_lib.odw_create(0, C.byref(ctx))
  if rc == 0:
    native_lib.odw_destroy(ctx)
  else:
    assert rc == 2
    assert b'HIP' in native_lib.odw_last_error(None) or native_lib.odw_last_error(None)


def test_product_never_imports_oracle():
  pkg = os.path.join(ROOT, 'freecad', 'optics_design_workbench_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h')):
        text = open(os.path.join(dirpath, f)).read()
        assert 'import oracle' not in text and 'from oracle' not in text and 'odw_oracle' not in text, f
