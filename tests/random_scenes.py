"""Random scenes for the randomised parity checks (tests/test_gpu_fuzz.py, tests/fuzz_parity.py):
primitives and booleans of random kind, size and placement, overlapping at random, in optical groups
of random type; all groups record, so whole trajectories are compared.  TEST INFRASTRUCTURE."""
import numpy as np

from freecad.optics_design_workbench_amd.freecad_elements import make
from freecad.optics_design_workbench_amd.scene import Document, bake


def rquat(rs):
  q = rs.normal(0, 1, 4)
  return tuple(q / np.linalg.norm(q))


def solid(doc, rs, k, centre, paraboloids=False):
  if paraboloids:
    kind = rs.choice(['box', 'sphere', 'cylinder', 'cone', 'torus', 'paraboloid'], p=[0.2, 0.2, 0.15, 0.1, 0.05, 0.3])
  else:
    kind = rs.choice(['box', 'sphere', 'cylinder', 'cone', 'torus'], p=[0.3, 0.3, 0.2, 0.1, 0.1])
  pl = dict(base=tuple(centre + rs.normal(0, 1.0, 3)), quat=rquat(rs))
  s = rs.uniform(2, 6)
  if kind == 'paraboloid':
    return make.makeParaboloid(doc, f'P{k}', rs.uniform(0.5, 6), rs.uniform(2, 8), **pl)
  if kind == 'box':
    return make.makeBox(doc, f'B{k}', *(rs.uniform(2, 8, 3)), **pl)
  if kind == 'sphere':
    return make.makeSphere(doc, f'S{k}', s, **pl)
  if kind == 'cylinder':
    return make.makeCylinder(doc, f'C{k}', s * 0.6, rs.uniform(3, 9), **pl)
  if kind == 'cone':
    return make.makeCone(doc, f'K{k}', s * 0.7, s * rs.uniform(0.0, 0.6), rs.uniform(3, 8), **pl)
  return make.makeTorus(doc, f'T{k}', s, s * rs.uniform(0.15, 0.4), **pl)


def scene(rs, rich=False, crowded=False, paraboloids=False):
  """rich: also tessellated solids (triangle primitives, BVH kernels), stochastic surfaces,
  gratings, absorbing media, partly reflecting mirrors and sequential mode;
  crowded: 8-20 groups (more than 16 primitives: grid / BVH kernels on analytic primitives with
  trimming conditions) and the distance tolerances the reference's documents use (1e-6 ... 1e-2);
  paraboloids: solid paraboloids among the primitives (off by default: the other modes keep the
  scenes their seeds have always produced)"""
  doc = Document()
  targets = []
  k = 0
  groups = []
  for g in range(rs.randint(8, 21) if crowded else rs.randint(2, 6)):
    centre = rs.uniform(-18, 18, 3)
    targets.append(centre)
    a = solid(doc, rs, k, centre, paraboloids); k += 1
    r = rs.rand()
    if rich and r < 0.2:
      elem = make.makeTessellated(doc, a, int(rs.choice([6, 12, 20])), smooth=bool(rs.rand() < 0.6))
    elif r < 0.45:
      elem = a
    else:
      b = solid(doc, rs, k, centre + rs.normal(0, 1.5, 3), paraboloids); k += 1
      elem = (make.makeCommon(doc, [a, b], f'X{k}') if r < 0.65 else
              make.makeCut(doc, a, b, f'X{k}') if r < 0.85 else make.makeFuse(doc, [a, b], f'X{k}'))
    kind = rs.choice(['Mirror', 'Lens', 'Absorber', 'Vacuum'], p=[0.3, 0.4, 0.15, 0.15])
    props = dict(RefractiveIndex=float(rs.uniform(1.2, 2.0))) if kind == 'Lens' else {}
    if rich:
      if kind == 'Mirror' and rs.rand() < 0.3:
        kind, props = 'Grating', dict(GratingType='Reflection', GratingLinesPerMillimeter=float(rs.uniform(200, 900)),
                                      GratingDiffractionOrder=int(rs.choice([-1, 1, 2])),
                                      GratingLinesOrientation=tuple(rs.normal(0, 1, 3)))
      elif kind == 'Mirror':
        props = dict(Reflectivity=float(rs.uniform(0.3, 1.0)))
        if rs.rand() < 0.5:
          props['ReflectedProbabilityDensity'] = str(rs.choice(['cos(theta-theta_refl)**8', 'exp(-(theta-theta_refl)**2/0.02)']))
        if rs.rand() < 0.3:
          props['RayModificationProbabilityDensity'] = 'exp(-theta**2/0.001)'
      elif kind == 'Lens':
        if rs.rand() < 0.4:
          props['AbsorptionLength'] = float(rs.uniform(2, 30))
        if rs.rand() < 0.3:
          props['RefractedProbabilityDensity'] = 'exp(-(theta-theta_refl)**2/0.005)'
    groups.append(make.makeOpticalGroup(doc, kind, [elem], **props))
  settings = dict(MaxIntersections=float(rs.choice([6, 12, 30])))
  if crowded:
    settings['DistanceTolerance'] = str(rs.choice(['1e-6', '1e-4', '1e-2']))
  if rich and rs.rand() < 0.25:
    settings['SequentialMode'] = True
    for step, i in enumerate(rs.permutation(len(groups))):
      settings[f'SequentialModeElements_{step:02d}'] = [groups[i]]
  make.makeSimulationSettings(doc, **settings)
  src = make.makePointSource(doc)
  sc = bake.bakeScene(doc, src)
  sc.group_record = np.ones_like(sc.group_record)
  return sc, bake.bakeLimits(doc, src), np.array(targets)


def rays(rs, targets, n, radius=60.0, spread=2.5):
  o = rs.normal(0, 1, (n, 3))
  o = o / np.linalg.norm(o, axis=1)[:, None] * radius
  t = targets[rs.randint(0, len(targets), n)] + rs.normal(0, spread, (n, 3))
  d = t - o
  return o, d / np.linalg.norm(d, axis=1)[:, None]
