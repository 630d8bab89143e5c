#!/usr/bin/env python3
"""Randomised parity: random scenes of primitives and booleans (random kinds, sizes, placements,
optical types, overlaps), random rays aimed at them, device vs oracle on whole trajectories
(all groups record).  Prints one JSON line per scene that differs and a summary.
  python tests/fuzz_parity.py [scenes] [rays] [seed] [rich]
(rich = 1: also tessellated solids, stochastic surfaces, gratings, absorbing media, sequential mode;
 3: crowded scenes of 8-20 groups with the reference's distance tolerances; 2: both;
 4: paraboloids among the primitives; 5: crowded with paraboloids)
Differences are classified: `tags` (a different sequence of hits: a real disagreement unless the
scene is chaotic -- many bounces between curved mirrors amplify rounding) and `coords` (same hits,
coordinates apart by more than 1e-7 mm).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

sys.path.insert(0, os.path.join(ROOT, 'tests'))
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
from oracle import capi as oracle        # this script is a checker, like the tests
from random_scenes import rays, scene

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rich = len(sys.argv) > 4 and sys.argv[4] in ('1', '2')
crowded = len(sys.argv) > 4 and sys.argv[4] in ('2', '3', '5')
parab = len(sys.argv) > 4 and sys.argv[4] in ('4', '5')


bad = dict(tags=0, coords=0)
done = 0
with Tracer(0) as tr:
  for s in range(n_scenes):
    rs = np.random.RandomState(seed0 * 100003 + s)
    try:
      sc, lim, targets = scene(rs, rich, crowded, parab)
    except Exception as e:                         # nested disjunctions etc.: not a parity matter
      continue
    o, d = rays(rs, targets, n_rays)
    tr.setScene(sc); tr.setLimits(lim); tr.setDetector(None)
    tr.reserveHits(n_rays * (lim.max_intersections + 1))
    tr.reset()
    tr.setSurfaceSeed(s + 17)
    tr.traceRays(o, d)
    tr.sync()
    g, gc = tr.hits(), tr.counters()
    ref = oracle.trace_rays(sc, lim, o, d, nthreads=0, surface_seed=s + 17)
    r = ref['hits']
    done += 1
    if done % 10 == 0:
      print(json.dumps(dict(progress=done)), flush=True)
    same_tags = len(g) == len(r) and np.array_equal(g['tag'], r['tag'])
    if not same_tags:
      # rays whose hit sequences differ
      gr = (g['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
      rr = (r['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
      cg, cr = np.bincount(gr, minlength=n_rays), np.bincount(rr, minlength=n_rays)
      which = np.flatnonzero(cg != cr)
      bad['tags'] += 1
      print(json.dumps(dict(scene=s, kind='tags', rays_differing=int(len(which)), first=which[:5].tolist(),
                            prims=[int(x) for x in sc.prim_type], counters_gpu=gc, counters_ref=ref['counters'])), flush=True)
      continue
    if len(g):
      dp = np.abs(g['point'] - r['point']).max()
      if dp > 1e-7:
        # the worst ray's deviations hit by hit: rounding amplified along a path of many bounces
        # (curved mirrors, grazing incidence) grows geometrically from ~1e-15; a disagreement of the
        # two implementations would start large
        dev = np.abs(g['point'] - r['point']).max(axis=1)
        ray = (g['tag'] & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64)
        worst = ray[np.argmax(dev)]
        seq = dev[ray == worst]
        first_big = dev[(dev > 1e-9)]
        # smallest deviation among the first hits (per ray) that exceed 1e-9
        firsts = []
        for rr_ in np.unique(ray[dev > 1e-9])[:200]:
          dd = dev[ray == rr_]
          firsts.append(float(dd[np.argmax(dd > 1e-9)]))
        bad['coords'] += 1
        print(json.dumps(dict(scene=s, kind='coords', max_dp=float(dp), prims=[int(x) for x in sc.prim_type],
                              worst_ray_deviations=[float('%.2e' % v) for v in seq[:14]],
                              largest_first_excess=max(firsts) if firsts else 0.0,
                              rays_above_1e9=int(len(np.unique(ray[dev > 1e-9]))))), flush=True)
print(json.dumps(dict(scenes=done, rays_each=n_rays, differing=bad)))
