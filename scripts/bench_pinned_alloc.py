#!/usr/bin/env python3
"""cost of page-locked host memory on this box: odw_host_alloc / odw_host_free of slabs of the run loop's size, and a
device-to-host copy into a fresh slab against one into a slab used before"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
tr = Tracer(0)
lib = tr._lib
for mb in (64, 300, 1200):
  size = mb << 20
  ts = []
  ptrs = []
  for k in range(5):
    p = C.c_void_p()
    t0 = time.perf_counter()
    tr._chk(lib.odw_host_alloc(tr._ctx, C.c_uint64(size), C.byref(p)), 'alloc')
    ts.append(time.perf_counter() - t0)
    ptrs.append(p)
  t0 = time.perf_counter()
  for p in ptrs:
    lib.odw_host_free(None, p)
  tf = (time.perf_counter() - t0) / len(ptrs)
  print(f'{mb:5d} MB  alloc {1e3 * np.median(ts):7.1f} ms (first {1e3 * ts[0]:.1f})   free {1e3 * tf:6.1f} ms', flush=True)
tr.close()
