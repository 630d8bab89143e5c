"""Global placement resolution (A1) against the reference's only exact pin:
test/22-global-placement/z-freecad-placements.py:40-70 lists the eight 4x4
matrices `allPlacementsAndPaths` must produce for the cube 'ShiftedCube',
which is reached through nested App::Parts, a DocumentObjectGroup, a LinkGroup
and App::Links with and without LinkTransform (scene: main.FCStd)."""
import os

import numpy as np

from conftest import SCENES

# expected translations (all rotations are identity), in the reference's order
EXPECTED = [(0, 0, -100), (3, 3, -100), (3, 0, -100), (3, -27, -100), (3, -27, -100),
            (3, 3, -97), (0, 0, -100), (0, -30, -100)]


def _doc():
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  return open_fcstd(os.path.join(SCENES, 'global-placement-main.FCStd'))


def test_shifted_cube_placements():
  from freecad.optics_design_workbench_amd.scene import allPlacementsAndPaths
  doc = _doc()
  cube = doc.getObjectsByLabel('ShiftedCube')[0]
  got = allPlacementsAndPaths(doc, cube)
  assert len(got) == 8
  for pl, _ in got:
    assert np.allclose(pl.Rotation, np.eye(3))
  bases = sorted(tuple(np.round(pl.Base, 9) + 0.0) for pl, _ in got)
  assert bases == sorted(tuple(float(v) for v in e) for e in EXPECTED)
  # ignoring links leaves only the container path
  only = allPlacementsAndPaths(doc, cube, ignoreLinks=True)
  assert len(only) == 1 and np.allclose(only[0][0].Base, [3, 3, -97])
  # paths are reported top-down and sorted lexically like the reference does
  paths = ['.'.join(p) for _, p in got]
  assert paths == sorted(paths) and all(p.endswith('Box003') for p in paths)


def test_proxy_repair_and_nested_bake():
  """main.FCStd was saved with Proxy = null; the reference re-attaches proxies
  from the property signature (common.py:181-242).  Groups nested in rotated
  Parts bake to the composed placements."""
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.scene import bake
  doc = _doc()
  assert [s.Name for s in bake.lightSources(doc)] == ['OpticalPointSource']
  assert [g._props['OpticalType'] for g in bake.opticalObjects(doc)] == ['Mirror', 'Lens', 'Mirror', 'Absorber']
  assert len(bake.simulationSettings(doc)) == 1
  pr = scenes.bakeProject(doc)
  # source: Part004 (z-3) > Part003 (z-12) > Part (x+5) > source (z+4)
  assert np.allclose(pr.source.xform.reshape(3, 4)[:, 3], [5, 0, -11])
  sc = pr.scene
  assert sc.n_prims == 4
  # lens sphere: Part002 o Part005 (z+9) o group (5,0,10) o sphere
  p2, p5, lg = doc.Part002.Placement, doc.Part005.Placement, doc.OpticalLensGroup.Placement
  assert np.allclose(sc.prim_to_world[1].Base, (p2 * p5 * lg).Base)
  # mirror cube: Part002 o mirror group o Part001 o Box
  exp = doc.Part002.Placement * doc.OpticalMirrorGroup.Placement * doc.Part001.Placement * doc.Box.Placement
  assert np.allclose(sc.prim_to_world[0].m, exp.m)


def test_nested_body_is_baked_from_its_brep_payload():
  """nested-structure.FCStd contains a PartDesign::Body (a padded hexagon: BRep only).  Its facets
  stand where the chain of placements puts the stored shape: every global placement of the mirror
  group o Part001 o Body.Placement o (shape in the body's own coordinates)"""
  from freecad.optics_design_workbench_amd.scene import bake, geometry, open_fcstd
  doc = open_fcstd(os.path.join(SCENES, 'nested-structure.FCStd'))
  old = geometry.BREP_EXACT
  geometry.BREP_EXACT = False          # as facets: their corners are the prism's (the exact form is eight half-spaces)
  try:
    sc = bake.bakeScene(doc, bake.lightSources(doc)[0])
  finally:
    geometry.BREP_EXACT = old
  g = list(sc.group_names).index('OpticalMirrorGroup001')
  tri = np.asarray(sc.prim_xform)[(sc.prim_group == g) & (sc.prim_type == 5)][:, :9].reshape(-1, 3)
  placements = bake.globalPlacements(doc, doc.getObject('OpticalMirrorGroup001'))
  assert len(placements) == 1 and len(tri) == 3 * 20 * len(placements)      # 20 facets of 3 corners per prism
  k = np.arange(6) * np.pi / 3
  hexagon = np.concatenate([np.stack([2 * np.cos(k), 2 * np.sin(k), np.full(6, z)], axis=1) for z in (0.0, 10.0)])
  want = []
  for pl in placements:
    m = (pl * doc.Part001.Placement * doc.Body.Placement).m
    want.append(hexagon @ m[:3, :3].T + m[:3, 3])
  want = np.concatenate(want)
  d = np.abs(tri[:, None, :] - want[None, :, :]).sum(axis=2).min(axis=1)
  assert d.max() < 1e-9                                   # every facet corner is a corner of a placed prism
  d = np.abs(want[:, None, :] - tri[None, :, :]).sum(axis=2).min(axis=1)
  assert d.max() < 1e-9                                   # and every corner is used


def test_link_scale():
  """App::Link.Scale: solids shown through a scaled link are magnified about the link's origin;
  a different scale per axis, and optical groups behind a scaled link, are refused"""
  import pytest
  from freecad.optics_design_workbench_amd.freecad_elements import make
  from freecad.optics_design_workbench_amd.scene import Document, Placement, bake, geometry
  doc = Document()
  ball = make.makeSphere(doc, 'S', 2.0, base=(1, 0, 0))
  box = make.makeBox(doc, 'B', 1, 2, 3, base=(0, 4, 0))
  both = make.makeCommon(doc, [ball, make.makeCylinder(doc, 'C', 1.5, 8, base=(1, 0, -4))], 'X')
  for target, check in ((ball, lambda n: n.kind == geometry.SPHERE and n.params[0] == 5.0 and np.allclose(n.placement.Base, (12.5, 20, 30))),
                        (box, lambda n: n.params[:3] == (2.5, 5.0, 7.5) and np.allclose(n.placement.Base, (10, 30, 30))),
                        (both, lambda n: n.op == 'common' and n.children[0].params[0] == 5.0
                         and np.allclose(n.children[1].placement.Base, (2.5, 0, -10)) and n.children[1].params[:2] == (3.75, 20.0))):
    link = doc.addObject('App::Link', 'L', LinkedObject=target, LinkTransform=True, Scale=2.5, ScaleVector=np.array([2.5, 2.5, 2.5]),
                         Placement=Placement(base=(10, 20, 30)))
    node, = geometry.solids_of(link)
    assert check(node), node
  link.ScaleVector = np.array([1.0, 2.0, 1.0])
  with pytest.raises(geometry.UnsupportedGeometry, match='per axis'):
    geometry.solids_of(link)
  link.ScaleVector = np.array([2.0, 2.0, 2.0])
  part = doc.addObject('App::Part', 'P', Group=[])
  grp = make.makeMirror(doc, [box])
  part.Group = [grp]
  doc.addObject('App::Link', 'LP', LinkedObject=part, Scale=2.0, ScaleVector=np.array([2.0, 2.0, 2.0]))
  with pytest.raises(geometry.UnsupportedGeometry, match='scaled link'):
    bake.allPlacementsAndPaths(doc, grp)
