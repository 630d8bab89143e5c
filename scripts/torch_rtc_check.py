#!/usr/bin/env python3
"""in a process that loaded torch first the run-time compiler is the one torch bundles (another ROCm release than
the system one this library was built with): the compiled kernel must still reproduce the generic one bit for bit"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
torch.cuda.init()
x = torch.ones(4, device='cuda') * 2          # torch's HIP runtime is up
import numpy as np
os.environ['ODW_KERNEL_CACHE'] = ''
from conftest import project
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
proj = project('lensesAndMirrors')
out = {}
for mode in ('off', 'structure'):
  with Tracer(0) as tr:
    tr.setScene(proj.scene); tr.setSource(proj.source); tr.setLimits(proj.limits); tr.setDetector(None)
    info = tr.compileScene(mode)
    tr.reserveHits(400000)
    tr.trace(0, 200000, 7); tr.sync()
    out[mode] = (tr.hits(), info)
a, b = out['off'][0], out['structure'][0]
print(out['structure'][1], all(np.array_equal(a[c], b[c]) for c in ('point', 'direction', 'power', 'tag')))
import subprocess
print([l.split()[-1] for l in open('/proc/self/maps') if 'hiprtc' in l or 'libamdhip64' in l or 'comgr' in l][:6])
