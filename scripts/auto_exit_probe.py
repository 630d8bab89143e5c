#!/usr/bin/env python3
"""a process that ends while the background compilation of ODW_COMPILE_AUTO is under way must end cleanly"""
import os, sys, tempfile, time
os.environ['ODW_KERNEL_CACHE'] = tempfile.mkdtemp(prefix='odw_kc_')
os.environ['ODW_SPEC_HOT_RAYS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
pr = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd'))
tr = Tracer(0)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits)
tr.compileScene('auto')
tr.reserveHits(200000)
for k in range(3):
  tr.reset(); tr.trace(0, 100000, 1); tr.sync()
print('mode', tr.compiledInfo(), 'exiting at once', flush=True)
t0 = time.time()
import atexit
atexit.register(lambda: print('python atexit after %.2f s' % (time.time() - t0), flush=True))
