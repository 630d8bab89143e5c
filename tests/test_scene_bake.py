"""FCStd-lite loader + scene bake against the facts SURVEY 8 lists for the
BASELINE scenes (object counts, placements, optical tables, sequences)."""
import numpy as np
import pytest

from conftest import project


def centre(scene, p):
  return scene.prim_to_world[p] * np.zeros(3)


def test_minimal():
  pr = project('minimal')
  sc = pr.scene
  assert sc.n_prims == 1 and sc.n_faces == 6 and sc.n_groups == 1
  assert sc.group_labels == ['OpticalAbsorberGroup'] and sc.group_type[0] == 3 and sc.group_record[0] == 1
  assert np.allclose(centre(sc, 0), [-5, -5, 15])          # group placed at (-5,-5,15)
  assert pr.limits.max_ray_length == pytest.approx(460.1823980554802)
  assert pr.limits.max_intersections == 100 and pr.limits.dist_tol == 1e-6 and pr.limits.power_tol == 1e-6
  assert pr.source.focal_length == 0 and pr.source.wavelength == 500
  assert np.allclose(pr.source.xform, [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0])


def test_lenses_and_mirrors():
  sc = project('lensesAndMirrors').scene
  assert sc.n_prims == 8 and sc.n_faces == 25            # 6+3+3+6+1+6 faces
  assert len(np.unique(sc.prim_solid)) == 6               # 6 shells
  assert sc.group_labels == ['OpticalMirrorGroup', 'OpticalLensGroup', 'OpticalMirrorGroup001',
                             'OpticalAbsorberGroup']
  assert list(sc.group_type) == [0, 1, 0, 3]
  assert list(sc.group_record) == [0, 0, 0, 1]
  assert np.allclose(sc.group_ior, 2.0) and np.allclose(sc.group_refl, 1.0) and np.all(np.isinf(sc.group_abslen))
  # Common(sphere R5 @z-4, cylinder R2 H10): each operand trimmed by the other
  assert list(sc.prim_type[1:5]) == [1, 2, 1, 2]
  conds = [(int(sc.cond_prim[c]), int(sc.cond_inside[c])) for c in range(len(sc.cond_prim))]
  assert conds == [(2, 1), (1, 1), (4, 1), (3, 1)]
  # cylinder top cap (z=10) can never be inside the sphere: pruned
  assert (sc.prim_flags[2] >> 8) == 0b011 and (sc.prim_flags[4] >> 8) == 0b011
  # group (-24,0,32) rot 90deg about y; sphere at z=-4 inside
  assert np.allclose(centre(sc, 1), [-28, 0, 32])
  # App::Link (0,0,-10) rot 180deg about y, LinkTransform=false
  assert np.allclose(centre(sc, 3), [-30, 0, 32])
  assert np.allclose(centre(sc, 4), [-34, 0, 32])
  # linked lens is the mirror image: its cylinder axis points to -x (global)
  axis = sc.prim_to_world[4].m[:3, :3] @ np.array([0, 0, 1.0])
  assert np.allclose(axis, [-1, 0, 0])
  axis0 = sc.prim_to_world[2].m[:3, :3] @ np.array([0, 0, 1.0])
  assert np.allclose(axis0, [1, 0, 0])
  assert sc.prim_type[6] == 4 and np.allclose(sc.prim_params[6][:2], [10, 2])
  assert np.allclose(centre(sc, 6), [-70, 0, 67])
  assert sc.seq_enabled == 0 and len(sc.seq_mask) == 0


def test_sequential_scene():
  sc = project('lensesAndMirrorsSequential').scene
  assert sc.seq_enabled == 1
  # [Mirror],[Lens],[Lens],[Mirror001],[Absorber]
  assert [int(m) for m in sc.seq_mask] == [1, 2, 2, 4, 8]


def test_huge_array():
  pr = project('hugeArray')
  sc = pr.scene
  assert sc.n_prims == 1500 and sc.n_faces == 1500 and sc.n_groups == 3
  assert list(sc.group_type) == [0, 1, 3]
  assert np.all(sc.prim_type == 1) and np.allclose(sc.prim_params[:, 0], 1.0)
  c = np.array([centre(sc, p) for p in range(1500)])
  for g, z0 in ((0, 26.0), (1, 0.0), (2, 51.0)):
    cg = c[sc.prim_group == g]
    assert len(cg) == 500
    assert np.allclose(cg.min(0), [-23, -23, z0]) and np.allclose(cg.max(0), [22, 22, z0 + 20])
    # Draft ortho array index order: z fastest, then y, then x; pitch 5
    assert np.allclose(cg[1] - cg[0], [0, 0, 5]) and np.allclose(cg[5] - cg[0], [0, 5, 0])
    assert np.allclose(cg[50] - cg[0], [5, 0, 0])
  # source at (-15,11,-18), rotated 11 deg about (1.1,1,1)
  m = pr.source.xform.reshape(3, 4)
  assert np.allclose(m[:, 3], [-15, 11, -18])
  ax = np.array([1.1, 1, 1]) / np.linalg.norm([1.1, 1, 1])
  assert np.allclose(m[:, :3] @ ax, ax)
  assert np.degrees(np.arccos((np.trace(m[:, :3]) - 1) / 2)) == pytest.approx(11.0, abs=1e-3)


def test_getting_started_and_cut():
  sc = project('GettingStarted').scene
  assert sc.n_prims == 4 and sc.n_faces == 15            # 6 + (sphere cap, cylinder, disc) + 6
  assert list(sc.group_type) == [0, 1, 3]
  # test/70 scene: absorber = Cut(Box 10x10x5 @z98, Sphere R100): the sphere's
  # face belongs to the result with its normal flipped
  s70 = project('source-and-absorber').scene
  assert list(s70.prim_type) == [0, 1]
  assert s70.prim_flags[1] & 1 == 1 and s70.prim_flags[0] & 1 == 0
  assert (int(s70.cond_prim[0]), int(s70.cond_inside[0])) == (1, 0)   # box faces outside the sphere
  assert (int(s70.cond_prim[1]), int(s70.cond_inside[1])) == (0, 1)   # sphere face inside the box
  assert np.isinf(project('source-and-absorber').source.focal_length)


def test_property_edit_rebakes():
  """doc.Sphere.Radius = r  (examples/1-getting-started/optimize-spotsize.ipynb cell 9)"""
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from conftest import SCENES
  import os
  doc = open_fcstd(os.path.join(SCENES, 'GettingStarted.FCStd'))
  rev = doc._revision
  doc.Sphere.Radius = 10.5
  assert doc._revision > rev
  sc = scenes.bakeProject(doc).scene
  assert sc.prim_params[2][0] == 10.5


def test_placement_algebra():
  from freecad.optics_design_workbench_amd.scene.placement import Placement, from_axis_angle
  a = Placement(base=(1, 2, 3), quat=(0, 0.3826834323650898, 0, 0.9238795325112867))   # 45 deg about y
  assert np.allclose(a.Rotation, from_axis_angle((0, 1, 0), np.pi / 4))
  assert np.allclose((a * a.inverse()).m, np.eye(4))
  assert np.allclose(a * np.array([1.0, 0, 0]), [1 + np.sqrt(.5), 2, 3 - np.sqrt(.5)])
  assert np.allclose(a.rows12().reshape(3, 4), a.m[:3])


def test_unsupported_geometry_is_loud():
  from freecad.optics_design_workbench_amd.scene import Document, UnsupportedGeometry
  from freecad.optics_design_workbench_amd.scene import geometry
  doc = Document()
  o = doc.addObject('Part::Sphere', 'S', Radius=1.0, Angle1=-90.0, Angle2=90.0, Angle3=270.0)
  with pytest.raises(UnsupportedGeometry, match='half a turn'):
    geometry.solids_of(o)                           # (more than half a turn: a disjunction of half-spaces)
  h = doc.addObject('Part::Sphere', 'H', Radius=1.0, Angle1=-90.0, Angle2=90.0, Angle3=180.0)
  node, = geometry.solids_of(h)                     # (half a turn and less: exact CSG since round 5)
  assert node.op == 'common' and [c.kind for c in node.children] == [geometry.SPHERE, geometry.BOX]
  f = doc.addObject('Part::Feature', 'Imported')
  with pytest.raises(UnsupportedGeometry):
    geometry.solids_of(f)
