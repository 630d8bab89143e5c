#!/usr/bin/env python3
"""what a notebook pays for the histogram of a run: files read back into host arrays (the reference's way) against the
rows kept in HBM (runSimulation(keepOnDevice=True), RawFolder.loadHits(device=True)).  python scripts/bench_keep_on_device.py [rays]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from freecad.optics_design_workbench_amd.jupyter_utils import FreecadDocument
n = float(sys.argv[1]) if len(sys.argv) > 1 else 5e7
tmp = tempfile.mkdtemp(prefix='odw_keep_', dir='/tmp')
path = os.path.join(tmp, 'lensesAndMirrors.FCStd')
shutil.copy(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd'), path)
kw = dict(bins=[np.linspace(-.05, .05, 200), np.linspace(-.05, .05, 200)])
with FreecadDocument(path) as f:
  f.OpticalSimulationSettings.EndAfterRays = '%g' % n
  f.runSimulation('true', compileScene='structure')                     # (warm-up: kernel compiled, buffers allocated)
  for keep in (False, True):
    t0 = time.perf_counter()
    raw = f.runSimulation('true', compileScene='structure', keepOnDevice=keep)
    t1 = time.perf_counter()
    hits = raw.loadHits(device=keep)
    t2 = time.perf_counter()
    H = hits.histogram(**kw)
    t3 = time.perf_counter()
    print(json.dumps(dict(keepOnDevice=keep, rays=n, hits=len(hits), run_s=round(t1 - t0, 3), loadHits_s=round(t2 - t1, 3),
                          histogram_s=round(t3 - t2, 3), counted=int(H.hist.sum()), kind=type(hits).__name__)), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
