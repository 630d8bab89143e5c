"""Point membership of a document's solids, straight from the boolean features.

TEST INFRASTRUCTURE: a second, independent reading of the geometry -- no faces, no trimming
conditions, no face masks, no convexity flags.  `scene/geometry.py` turns a boolean tree into
primitive FACES with conjunctions of inside / outside tests (what the tracer needs); this
module answers only "is the point inside the solid?" by evaluating the features as they are
written in `Document.xml`:
    Part::Box / Sphere / Cylinder / Cone / Torus     the point in the primitive's frame
    Part::MultiCommon / Common                        all operands
    Part::MultiFuse / Fuse                            any operand
    Part::Cut                                         Base and not Tool
    App::Link (LinkTransform false / true)            the target, its own placement dropped / kept
    Draft Array (PlacementList)                       one solid per listed placement
    App::Part / LinkGroup / Compound / Group          their members
The oracle and the device share the bake (tests compare them with each other); a wrong trimming
rule, normal flip or face mask in the bake is invisible to those tests and visible here
(tests/test_bake_independent.py).
"""
import numpy as np

from freecad.optics_design_workbench_amd.scene.placement import Placement


def _own(obj, keep=True):
  return obj.Placement if (keep and obj.hasProperty('Placement')) else Placement.identity()


def _local(pl, pts):
  inv = pl.inverse().m
  return pts @ inv[:3, :3].T + inv[:3, 3]


class Solid:
  """one solid (shell) of an optical group: membership of world points"""

  def __init__(self, fn, name):
    self._fn, self.name = fn, name

  def inside(self, pts):
    return self._fn(np.asarray(pts, dtype=np.float64).reshape(-1, 3))


def _primitive(obj):
  t = obj.TypeId
  if t == 'Part::Box':
    L = np.array([obj.Length, obj.Width, obj.Height], dtype=np.float64)
    return lambda q: np.all((q >= 0) & (q <= L), axis=1)
  if t == 'Part::Sphere':
    R = float(obj.Radius)
    return lambda q: (q**2).sum(axis=1) <= R * R
  if t == 'Part::Cylinder':
    R, H = float(obj.Radius), float(obj.Height)
    return lambda q: (q[:, 0]**2 + q[:, 1]**2 <= R * R) & (q[:, 2] >= 0) & (q[:, 2] <= H)
  if t == 'Part::Cone':
    R1, R2, H = float(obj.Radius1), float(obj.Radius2), float(obj.Height)
    return lambda q: (np.hypot(q[:, 0], q[:, 1]) <= R1 + (R2 - R1) * q[:, 2] / H) & (q[:, 2] >= 0) & (q[:, 2] <= H)
  if t == 'Part::Torus':
    R1, R2 = float(obj.Radius1), float(obj.Radius2)
    return lambda q: (np.hypot(q[:, 0], q[:, 1]) - R1)**2 + q[:, 2]**2 <= R2 * R2
  if t == 'Part::FeaturePython' and obj.ProxyClass == 'Paraboloid':
    f, H = float(obj.FocalLength), float(obj.Height)
    return lambda q: (q[:, 0]**2 + q[:, 1]**2 <= 4 * f * q[:, 2]) & (q[:, 2] <= H)
  return None


def _is_array(obj):
  return obj.ProxyClass == 'Array' and (obj.ProxyModule or '').startswith('draftobjects')


def members(obj, keep_placement=True):
  """-> list of membership functions (points in the coordinates of obj's container), one per solid"""
  own = _own(obj, keep_placement)
  t = obj.TypeId
  prim = _primitive(obj)
  if prim is not None:
    return [lambda p, f=prim, pl=own: f(_local(pl, p))]

  def single(child):
    fs = members(child)
    assert len(fs) == 1, f'{obj.Name}: boolean operand {child.Name} is not a single solid'
    return fs[0]
  if t in ('Part::MultiCommon', 'Part::MultiFuse', 'Part::Common', 'Part::Fuse', 'Part::Cut'):
    kids = [single(c) for c in (obj.Shapes if t.startswith('Part::Multi') else (obj.Base, obj.Tool))]
    if t in ('Part::MultiCommon', 'Part::Common'):
      comb = lambda q: np.logical_and.reduce([k(q) for k in kids])
    elif t in ('Part::MultiFuse', 'Part::Fuse'):
      comb = lambda q: np.logical_or.reduce([k(q) for k in kids])
    else:
      comb = lambda q: kids[0](q) & ~kids[1](q)
    return [lambda p, f=comb, pl=own: f(_local(pl, p))]
  if _is_array(obj):
    # (a Draft array made with links: the base object's own placement does not count)
    use_link = bool((obj._props.get('Proxy') or {}).get('state', {}).get('use_link', True))
    base = members(obj.Base, keep_placement=not use_link or bool(obj._props.get('LinkTransform', False)))
    count = int(obj._props.get('Count', len(obj.PlacementList)))
    out = []
    for pl in obj.PlacementList[:count]:
      for f in base:
        out.append(lambda p, f=f, a=own * pl: f(_local(a, p)))
    return out
  if t.startswith('App::Link') and not t.startswith('App::LinkGroup'):
    target = obj.LinkedObject
    base = members(target, keep_placement=bool(obj._props.get('LinkTransform', False)))
    return [lambda p, f=f, pl=own: f(_local(pl, p)) for f in base]
  if t.startswith('App::LinkGroup') or t in ('App::Part', 'App::DocumentObjectGroup', 'Part::Compound'):
    key = 'ElementList' if t.startswith('App::LinkGroup') else ('Links' if t == 'Part::Compound' else 'Group')
    out = []
    for c in obj._props.get(key) or []:
      out += [lambda p, f=f, pl=own: f(_local(pl, p)) for f in members(c)]
    return out
  raise NotImplementedError(f'{obj.Name}: {t}')


def _containers(doc, obj):
  out = []
  for o in doc.Objects:
    for key in ('Group', 'ElementList'):
      if obj in (o._props.get(key) or []) and o is not obj:
        out.append(o)
  return out


def globalPlacement(doc, obj):
  """product of the placements from the top level down to obj (single-instance documents)"""
  cs = _containers(doc, obj)
  assert len(cs) <= 1, f'{obj.Name} sits in several containers'
  own = _own(obj)
  return own if not cs else globalPlacement(doc, cs[0]) * own


def groupSolids(doc):
  """{optical group name: [Solid, ...]} in the order the members are listed"""
  out = {}
  for g in doc.Objects:
    if g.TypeId == 'App::LinkGroupPython' and g.ProxyClass == 'OpticalGroupProxy':
      gp = globalPlacement(doc, g)
      solids = []
      for child in g._props.get('ElementList') or []:
        for k, f in enumerate(members(child)):
          solids.append(Solid(lambda p, f=f, pl=gp: f(_local(pl, p)), f'{g.Name}/{child.Name}[{k}]'))
      out[g.Name] = solids
  return out
