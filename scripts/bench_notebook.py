#!/usr/bin/env python3
"""End-to-end latency of the notebook workflow of
examples/1-getting-started/optimize-spotsize.ipynb: 30 x (set a lens radius,
runSimulation('true') with EndAfterRays = 1e3, load the hits, histogram).
The reference's recorded run of this loop took 12 min 43 s.
  python scripts/bench_notebook.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument

t0 = time.perf_counter()
with FreecadDocument(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'GettingStarted.FCStd'), workInTempCopy=True) as f:
  f.OpticalSimulationSettings.EndAfterRays = '1e3'
  t_open = time.perf_counter() - t0
  times, sizes = [], []
  for r in np.linspace(9, 11, 30):
    t1 = time.perf_counter()
    f.Sphere.Radius = float(r)
    raw = f.runSimulation('true')
    H = raw.loadHits('*').histogram(bins=30)
    sizes.append(float(H.hist.max()))
    times.append(time.perf_counter() - t1)
print(json.dumps(dict(simulations=30, rays_each=1000, open_s=round(t_open, 3), first_s=round(times[0], 3),
                      median_s=round(float(np.median(times)), 4), total_s=round(time.perf_counter() - t0, 3),
                      reference_total_s=763)))
