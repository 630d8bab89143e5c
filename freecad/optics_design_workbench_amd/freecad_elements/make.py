"""Create optical elements in a document programmatically.

Counterpart of the reference's toolbar commands (optical_group.py:364-403
`AddOpticalGroup`, point_source.py:690-710 `AddPointSource`,
simulation_settings.py `AddSimulationSettings`): new objects get the same
property names and defaults as the reference's `_properties()` tables
(optical_group.py:29-96, point_source.py:32-70, generic_source.py:23-37,
simulation_settings.py:20-77), so documents built here bake like loaded ones.
"""
import numpy as np

from ..scene.placement import Placement

_GROUP_DEFAULTS = dict(
    RefractiveIndex=2.0, ReflectedProbabilityDensity='', RefractedProbabilityDensity='',
    PowerThetaDomain='-pi/2, pi/2', PowerPhiDomain='0, 2*pi', RayModificationProbabilityDensity='',
    ModifyThetaDomain='-pi/2, pi/2', ModifyPhiDomain='0, 2*pi', Reflectivity=1.0,
    AbsorptionLength='inf', GratingType='Reflection', GratingLinesPerMillimeter=1000.0,
    GratingLinesOrientation=np.array([0.0, 0.0, 1.0]), GratingDiffractionOrder=1)

_RECORD_DEFAULT = dict(Mirror=False, Lens=False, Grating=False, Absorber=True, Vacuum=True)

_SOURCE_DEFAULTS = dict(
    PowerDensity='exp(-theta^2/0.01)', Wavelength=500.0, FocalLength='0', Divergence='-',
    ThetaDomain='0, pi/4', PhiDomain='0, 2*pi', RadiusDomain='0, 10', RandomNumberGeneratorMode='?',
    ThetaResolutionNumericMode='1e5', RadiusResolutionNumericMode='1e5', PhiResolutionNumericMode='1e2',
    Fans=2, FanPhi0='0', RaysPerFan=20, FanModePowerSpan=0.9, RecordRays=False,
    IgnoredOpticalElements=[], RaysPerIterationScale=1.0, MaxIntersectionsScale=1.0, MaxRayLengthScale=1.0)

_SETTINGS_DEFAULTS = dict(
    Active=True, EnableStoreSingleShotData=False, EndAfterIterations='inf', EndAfterRays='1e4',
    EndAfterHits='inf', RaysPerIteration=100.0, MaxIntersections=100.0, DistanceTolerance='1e-6',
    MaxRayLength=1000.0, ShowRaysInContinuousMode=True, WorkerProcessCount='num_cpus',
    SequentialMode=False, SequentialModeElements_00=[],
    StoreHitInitPoint=False, StoreHitInitDirection=False, StoreHitInitPower=False, StoreHitInitWavelength=False,
    StoreHitInitPhi=False, StoreHitInitTheta=False, StoreHitRayIndex=False, StoreHitFanIndex=False,
    StoreHitTotalFanCount=False, StoreHitTotalRaysInFan=False)


def _placement(base=(0, 0, 0), quat=(0, 0, 0, 1), placement=None):
  return placement if placement is not None else Placement(base=base, quat=quat)


def makeBox(doc, name='Box', length=10.0, width=10.0, height=10.0, **pl):
  return doc.addObject('Part::Box', name, Length=float(length), Width=float(width), Height=float(height),
                       Placement=_placement(**pl))


def makeSphere(doc, name='Sphere', radius=5.0, **pl):
  return doc.addObject('Part::Sphere', name, Radius=float(radius), Angle1=-90.0, Angle2=90.0, Angle3=360.0,
                       Placement=_placement(**pl))


def makeCylinder(doc, name='Cylinder', radius=2.0, height=10.0, **pl):
  return doc.addObject('Part::Cylinder', name, Radius=float(radius), Height=float(height), Angle=360.0,
                       Placement=_placement(**pl))


def makeCone(doc, name='Cone', radius1=2.0, radius2=4.0, height=10.0, **pl):
  return doc.addObject('Part::Cone', name, Radius1=float(radius1), Radius2=float(radius2),
                       Height=float(height), Angle=360.0, Placement=_placement(**pl))


def makeTorus(doc, name='Torus', radius1=10.0, radius2=2.0, **pl):
  return doc.addObject('Part::Torus', name, Radius1=float(radius1), Radius2=float(radius2),
                       Angle1=-180.0, Angle2=180.0, Angle3=360.0, Placement=_placement(**pl))


def makeParaboloid(doc, name='Paraboloid', focalLength=10.0, height=5.0, **pl):
  """solid paraboloid of revolution x^2 + y^2 <= 4 f z, z <= height in its own frame (vertex at the
  origin, axis +z): the blank of a parabolic mirror.  FreeCAD has no such primitive (there it is the
  revolution of a parabola); this feature carries the two numbers the tracer needs"""
  return doc.addObject('Part::FeaturePython', name, Proxy={'module': 'freecad.optics_design_workbench_amd.scene.geometry',
                                                             'class': 'Paraboloid', 'state': {}},
                       FocalLength=float(focalLength), Height=float(height), Placement=_placement(**pl))


def makeCommon(doc, shapes, name='Common', **pl):
  return doc.addObject('Part::MultiCommon', name, Shapes=list(shapes), Placement=_placement(**pl))


def makeCut(doc, baseObject, toolObject, name='Cut', **pl):
  return doc.addObject('Part::Cut', name, Base=baseObject, Tool=toolObject, Placement=_placement(**pl))


def makeFuse(doc, shapes, name='Fusion', **pl):
  return doc.addObject('Part::MultiFuse', name, Shapes=list(shapes), Placement=_placement(**pl))


def makeOpticalGroup(doc, opticalType, elements, name=None, placement=None, **props):
  """`OpticalType` in Mirror|Lens|Grating|Absorber|Vacuum; RecordHits follows
  OpticalGroupProxy.onChanged (optical_group.py:141-160) unless given"""
  if opticalType not in _RECORD_DEFAULT:
    raise ValueError(f'invalid optical type {opticalType!r}')
  p = dict(_GROUP_DEFAULTS)
  p['RecordHits'] = _RECORD_DEFAULT[opticalType]
  p.update(props)
  return doc.addObject('App::LinkGroupPython', name or f'Optical{opticalType}Group',
                       Proxy={'module': 'freecad.optics_design_workbench.freecad_elements.optical_group',
                              'class': 'OpticalGroupProxy', 'state': {'oldType': opticalType}},
                       OpticalType=opticalType, ElementList=list(elements),
                       Placement=placement or Placement.identity(), **p)


def makeMirror(doc, elements, **kw):
  return makeOpticalGroup(doc, 'Mirror', elements, **kw)


def makeLens(doc, elements, **kw):
  return makeOpticalGroup(doc, 'Lens', elements, **kw)


def makeAbsorber(doc, elements, **kw):
  return makeOpticalGroup(doc, 'Absorber', elements, **kw)


def makeVacuum(doc, elements, **kw):
  return makeOpticalGroup(doc, 'Vacuum', elements, **kw)


def makeGrating(doc, elements, **kw):
  return makeOpticalGroup(doc, 'Grating', elements, **kw)


def makePointSource(doc, name='OpticalPointSource', placement=None, **props):
  p = dict(_SOURCE_DEFAULTS)
  p.update(props)
  return doc.addObject('App::LinkGroupPython', name,
                       Proxy={'module': 'freecad.optics_design_workbench.freecad_elements.point_source',
                              'class': 'PointSourceProxy', 'state': {}},
                       ElementList=[], Placement=placement or Placement.identity(), **p)


def makeSimulationSettings(doc, name='OpticalSimulationSettings', **props):
  p = dict(_SETTINGS_DEFAULTS)
  p.update(props)
  return doc.addObject('Part::FeaturePython', name,
                       Proxy={'module': 'freecad.optics_design_workbench.freecad_elements.simulation_settings',
                              'class': 'SimulationSettingsProxy', 'state': {}}, **p)


def makeReplaySource(doc, replayFromDir, name='OpticalReplaySource', placement=None, **props):
  """ReplaySourceProxy (replay_source.py:30-38): replays the hits below `replayFromDir` as rays"""
  p = dict(ReplayFromDir=str(replayFromDir), Wavelength=500.0, RecordRays=False, IgnoredOpticalElements=[],
           RaysPerIterationScale=1.0, MaxIntersectionsScale=1.0, MaxRayLengthScale=1.0)
  p.update(props)
  return doc.addObject('App::LinkGroupPython', name,
                       Proxy={'module': 'freecad.optics_design_workbench.freecad_elements.replay_source',
                              'class': 'ReplaySourceProxy', 'state': {}},
                       ElementList=[], Placement=placement or Placement.identity(), **p)


def makeMesh(doc, vertices, triangles, vertexNormals=None, name='Mesh', **pl):
  """a tessellated shape (what FreeCAD's `Shape.tessellate(tol)` or an STL export gives):
  vertices (n,3), triangles (m,3) counter-clockwise seen from outside, optional unit
  normals per vertex for smooth shading of curved faces"""
  props = dict(Vertices=np.asarray(vertices, dtype=np.float64).reshape(-1, 3),
               Triangles=np.asarray(triangles, dtype=np.int64).reshape(-1, 3), Placement=_placement(**pl))
  if vertexNormals is not None:
    props['VertexNormals'] = np.asarray(vertexNormals, dtype=np.float64).reshape(-1, 3)
  return doc.addObject('Mesh::Feature', name, **props)


def makeTessellated(doc, solid, segments=48, smooth=True, name=None):
  """mesh of a primitive solid object (Part::Sphere / Cylinder / Cone / Torus / Box) at the same placement"""
  from ..scene import geometry
  node = geometry._primitive_of(solid)
  if node is None:
    raise geometry.UnsupportedGeometry(f'{solid.Name}: only primitive solids can be tessellated without FreeCAD')
  v, tri, vn = geometry.tessellate(node.kind, node.params, segments)
  return makeMesh(doc, v, tri, vn if smooth else None, name=name or solid.Name + 'Mesh',
                  placement=solid.Placement)
