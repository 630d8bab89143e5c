#!/usr/bin/env python3
"""Where the time of a C3 launch goes: the same 1e8 rays with parts of the work
switched off through the public knobs (record flags, intersection cap).
  python scripts/ablate.py [n_rays]
"""
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from freecad.optics_design_workbench_amd import _native, scenes

if os.environ.get('ODW_VARIANT_LIB'):      # kernel experiments: a library built with other -D flags
  _native.LIB_PATH = os.path.abspath(os.environ['ODW_VARIANT_LIB'])
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
pr = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd'))
det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=1024, ny=1024, toward=pr.source.xform[[3, 7, 11]])
tr = Tracer(0)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(det)
tr.reserveHits(n + 1024)
tr.timingEnable(True)


def run(label, record_hits=True, histogram=True, cap=None, reps=3):
  lim = copy.copy(pr.limits)
  if cap is not None:
    lim.max_intersections = cap
  tr.setLimits(lim)
  best = 1e9
  for _ in range(reps):
    tr.reset()
    tr.timingRead()
    tr.trace(0, n, 0x0D15EA5E, record_hits=record_hits, histogram=histogram)
    tr.sync()
    ms, launches = tr.timingRead()
    best = min(best, ms)
  c = tr.counters()
  print(json.dumps(dict(case=label, ms=round(best, 3), segments_per_ray=c['segments'] / n,
                        hits_per_ray=c['recorded_hits'] / n)), flush=True)
  return best


run('full')
if os.environ.get('ODW_ABLATE_QUICK'):
  sys.exit(0)
run('no hit rows', record_hits=False)
run('no histogram', histogram=False)
run('no recording at all', record_hits=False, histogram=False)
for cap in range(0, 8):
  run(f'cap {cap} intersections, no recording', record_hits=False, histogram=False, cap=cap)
