#!/usr/bin/env python3
"""Randomised parity of the sampler path: random point sources (density expression, focal length
finite / negative / infinite, domains, placement) over random scenes, `odw_trace` with a random
detector window (any plane, all groups) and the segment list, device vs oracle.
  python tests/fuzz_sources.py [scenes] [rays] [seed]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))   # TEST INFRASTRUCTURE: a checker that runs the oracle next to the device
import numpy as np

from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import bake
from freecad.optics_design_workbench_amd.scene.placement import Placement
from freecad.optics_design_workbench_amd.simulation.simulation_loop import bakeLightSource
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from oracle import capi as oracle
import random_scenes

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
DENS = ['exp(-theta^2/0.01)', '1', 'cos(theta)', 'exp(-theta^2/0.3)*(1+0.5*cos(3*phi))', 'exp(-r^2/4)', 'r*exp(-r)*(1.2+sin(phi))']
bad = dict(counters=0, tags=0, hist=0, segs=0, coords=0)
done = 0
with Tracer(0) as tr:
  for s in range(n_scenes):
    if os.environ.get('ODW_FUZZ_ONLY') and s != int(os.environ['ODW_FUZZ_ONLY']):
      continue                                     # (one scene in detail: its differing rays segment by segment)
    rs = np.random.RandomState(seed0 * 7919 + s)
    doc_rs = np.random.RandomState(seed0 * 100003 + s)
    try:
      # (scene() makes its own source; build the document again with ours)
      sc0, lim0, targets = random_scenes.scene(doc_rs, rich=bool(s % 2))
    except Exception:
      continue
    # a fresh document with the same optics is not available from scene(): reuse its bake and swap the source
    from freecad.optics_design_workbench_amd.scene import Document
    doc = Document()
    make.makeSimulationSettings(doc)
    focal = rs.choice(['0', '25', '-40', 'inf'])
    radial = focal == 'inf'
    dens = DENS[rs.randint(4, 6)] if radial else DENS[rs.randint(0, 4)]
    target = targets[rs.randint(len(targets))]
    pos = rs.normal(0, 1, 3); pos = pos / np.linalg.norm(pos) * 55.0
    z = (target - pos) / np.linalg.norm(target - pos)
    x = np.cross(z, [0.3, 0.5, 0.8]); x /= np.linalg.norm(x)
    m = np.eye(4); m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, np.cross(z, x), z, pos
    props = dict(PowerDensity=dens, FocalLength=focal, ThetaDomain=f'0, {rs.uniform(0.05, 0.6):.3f}')
    if radial:
      props['RadiusDomain'] = f'0, {rs.uniform(2, 8):.2f}'
    src = make.makePointSource(doc, placement=Placement(matrix=m), **props)
    try:
      bs = bakeLightSource(doc, src, 0)
    except Exception as e:
      print(json.dumps(dict(scene=s, skip=str(e)[:80])), flush=True)
      continue
    sc, lim = sc0, lim0
    # detector: a random plane through a random target
    ez = rs.normal(0, 1, 3); ez /= np.linalg.norm(ez)
    ex = np.cross(ez, [1, 0, 0.2]); ex /= np.linalg.norm(ex)
    det = dict(group=-1, origin=target.tolist(), ex=ex.tolist(), ey=np.cross(ez, ex).tolist(),
               x_lo=-12.0, x_hi=12.0, y_lo=-9.0, y_hi=15.0, nx=64, ny=48)
    seed = 1000 + s
    first = int(rs.randint(0, 1 << 40))
    tr.setScene(sc); tr.setSource(bs); tr.setLimits(lim); tr.setDetector(det)
    tr.reserveHits(n * (lim.max_intersections + 1)); tr.reserveSegments(n * lim.max_intersections)
    tr.reset()
    if first + n > (1 << 40):
      first = 0
    tr.trace(first, n, seed, record_segments=True)
    tr.sync()
    g, gc, gh, gs = tr.hits(), tr.counters(), tr.histogram(), tr.segments()
    ref = oracle.trace(sc, bs, lim, first, n, seed, det=det, nthreads=0, hit_capacity=n * (lim.max_intersections + 1))
    rseg = oracle.trace_segments(sc, lim, src=bs, first=first, n=n, seed=seed)['segments']
    done += 1
    what = []
    if gc != ref['counters']:
      what.append('counters')
    if len(g) != len(ref['hits']) or not np.array_equal(g['tag'], ref['hits']['tag']):
      what.append('tags')
    elif len(g) and np.abs(g['point'] - ref['hits']['point']).max() > 1e-7:
      what.append('coords')
    if not np.array_equal(gh, ref['hist']):
      what.append('hist')
    if len(gs) != len(rseg) or not np.array_equal(gs['tag'], rseg['tag']):
      what.append('segs')
    for w in what:
      bad[w] += 1
    if [w for w in what if w != 'coords']:
      print(json.dumps(dict(scene=s, differs=what, focal=str(focal), density=dens, counters_gpu=gc, counters_ref=ref['counters'],
                            hist_diff=int(np.abs(gh.astype(np.int64) - ref['hist'].astype(np.int64)).sum()))), flush=True)
    if os.environ.get('ODW_FUZZ_ONLY') and 'segs' in what:
      ray_of = lambda a: (a['tag'] & np.uint64((1 << 40) - 1)).astype(np.int64)
      cg, cr = np.bincount(ray_of(gs) - first, minlength=n), np.bincount(ray_of(rseg) - first, minlength=n)
      print('prims', sc.prim_type.tolist(), 'groups', sc.prim_group.tolist(), 'group types', sc.group_type.tolist(),
            'cond', sc.prim_cond_off.tolist(), 'tol', lim.dist_tol)
      for i in np.flatnonzero(cg != cr)[:3]:
        for name, a in (('gpu', gs), ('oracle', rseg)):
          rows = a[ray_of(a) - first == i]
          print(name, 'ray', int(i), [(np.round(r['p1'], 6).tolist(), np.round(r['p2'], 6).tolist(), int(r['tag'] >> np.uint64(52))) for r in rows])
    if done % 10 == 0:
      print(json.dumps(dict(progress=done)), flush=True)
print(json.dumps(dict(scenes=done, rays_each=n, differing=bad)))
