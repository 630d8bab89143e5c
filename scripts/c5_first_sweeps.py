#!/usr/bin/env python3
"""why the first sweeps of a process are slower: the sweep's own timeline (ODW_SWEEP_TRACE) of sweeps 1 .. n, summed per kind of
step"""
import collections
import contextlib
import io
import os
import re
import sys
import time

os.environ['ODW_SWEEP_TRACE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import sweep
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

radii = np.linspace(9, 11, 64)
doc = open_fcstd(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'))
tr = Tracer(0)
doc.Sphere.Radius = float(radii[0])
first = scenes.bakeProject(doc)
tr.setScene(first.scene); tr.setLimits(first.limits)
tr.compileScene('structure')
traces = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
  buf = io.StringIO()
  tr.sync(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  with contextlib.redirect_stderr(buf):
    sweep.parameterSweep(doc, lambda d, r: setattr(d.Sphere, 'Radius', float(r)), radii, rays=10_000_000, seed=0x0D15EA5E, tracer=tr,
                         measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot), keepSample=1000)
  tr.sync(); torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) * 1e3
  kinds = collections.defaultdict(float)
  last = 0.0
  for line in buf.getvalue().splitlines():
    m = re.search(r"\s([a-z+]+)[ \[].*?(-?\d+\.\d+)\s+(-?\d+\.\d+)\s*$", line)
    if m:
      kinds[m.group(1)] += float(m.group(3)) - float(m.group(2))
      last = max(last, float(m.group(3)))
  traces.append((ms, buf.getvalue()))
  print(f'sweep {k}: {ms:6.1f} ms  last mark {last:6.1f}  ' + '  '.join(f'{a} {b:.1f}' for a, b in sorted(kinds.items())), flush=True)

later = traces[2:]
for label, (ms, text) in (('fastest', min(later)), ('slowest', max(later))):
  print(f'---- {label}: {ms:.1f} ms')
  print(text)
