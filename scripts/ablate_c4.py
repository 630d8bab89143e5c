#!/usr/bin/env python3
"""hugeArray launch with parts of the recording switched off (kernel ms by HIP events)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 125000000
pr = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'hugeArray.FCStd'))
gi = pr.scene.group_index('OpticalAbsorberGroup')
det = dict(group=gi, origin=[-0.5, -0.5, 61.0], ex=[1.0, 0.0, 0.0], ey=[0.0, 1.0, 0.0], x_lo=-25.0, x_hi=25.0, y_lo=-25.0, y_hi=25.0, nx=1024, ny=1024)
tr = Tracer(0)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(det)
tr.reserveHits(n // 2)
tr.timingEnable(True)
for label, rh, hg in (('full', True, True), ('no hit rows', False, True), ('no histogram', True, False), ('no recording', False, False)):
  best = 1e9
  for _ in range(4):
    tr.reset(); tr.timingRead()
    tr.trace(0, n, 0x0D15EA5E, record_hits=rh, histogram=hg)
    tr.sync()
    best = min(best, tr.timingRead()[0])
  print(json.dumps(dict(case=label, ms=round(best, 3))), flush=True)
tr.close()      # (diagnostic builds print their phase statistics from odw_destroy)
