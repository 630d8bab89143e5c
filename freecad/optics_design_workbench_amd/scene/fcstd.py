"""FCStd-lite: read a FreeCAD document without FreeCAD.

An .FCStd file is a zip; `Document.xml` lists every document object with its
typed properties.  This module parses exactly the property kinds the hot path
needs: parametric Part primitives, booleans, App::Link, Draft
link arrays, App::LinkGroupPython optical groups / light sources and the
simulation settings object.  Geometry is rebuilt from the parametric features,
which is what makes the GPU box (no FreeCAD) self-sufficient; the BRep
payloads (`*.brp`) are kept as bytes and read only for objects that have no
parametric recipe (`Part::Feature` imports, `PartDesign::Body`:
scene/brep.py, scene/brep_mesh.py).

It mirrors what the reference reaches through the FreeCAD API
(`obj.Placement`, `obj.Radius`, `group.ElementList`, `obj.Proxy` class names;
freecad_elements/find.py:59-141) -- objects expose their properties as
attributes, and setting an attribute marks the document dirty so the scene is
re-baked before the next simulation (jupyter_utils/freecad_document.py
property round trips).
"""
import base64
import json
import struct
import xml.etree.ElementTree as ET
import zipfile

import numpy as np

from .placement import Placement


class DocumentObject:
  """One FreeCAD document object: `Name`, `TypeId`, properties as attributes."""

  def __init__(self, document, name, type_id):
    object.__setattr__(self, '_doc', document)
    object.__setattr__(self, 'Name', name)
    object.__setattr__(self, 'TypeId', type_id)
    object.__setattr__(self, '_props', {})
    object.__setattr__(self, '_types', {})

  def __getattr__(self, key):
    props = object.__getattribute__(self, '_props')
    if key in props:
      return props[key]
    raise AttributeError(f'{self.Name} ({self.TypeId}) has no property {key!r}')

  def __setattr__(self, key, value):
    self._props[key] = value
    self._doc._touch()
    # a shape-defining property of an object with a stored shape changed: stored shapes of the
    # project may be out of date (placements are applied at bake time and do not count)
    if key not in ('Placement', 'Label', 'Visibility') and self._props.get('Shape') is not None:
      self._doc._shape_revision = getattr(self._doc, '_shape_revision', 0) + 1

  def hasProperty(self, key):
    return key in self._props

  @property
  def PropertiesList(self):
    return sorted(self._props)

  @property
  def ProxyClass(self):
    p = self._props.get('Proxy')
    return p.get('class') if isinstance(p, dict) else None

  @property
  def ProxyModule(self):
    p = self._props.get('Proxy')
    return p.get('module') if isinstance(p, dict) else None

  def isDerivedFrom(self, type_id):
    return self.TypeId == type_id or self.TypeId.startswith(type_id)

  def __repr__(self):
    return f'<{self.TypeId} {self.Name} ({self._props.get("Label", "")})>'


def _parse_enum(prop):
  integer = prop.find('Integer')
  idx = int(integer.attrib['value'])
  custom = prop.find('CustomEnumList')
  if custom is not None:
    names = [e.attrib['value'] for e in custom.findall('Enum')]
    if 0 <= idx < len(names):
      return names[idx]
  return idx


def _parse_property(prop, zf):
  ptype = prop.attrib['type'].replace('App::Property', '')
  child = next(iter(prop), None)
  if child is None:
    return None
  tag = child.tag
  if ptype == 'Enumeration':
    return _parse_enum(prop)
  if tag == 'Float':
    return float(child.attrib['value'])
  if tag == 'Integer':
    return int(child.attrib['value'])
  if tag == 'Bool':
    return child.attrib['value'].lower() == 'true'
  if tag == 'String':
    return child.attrib['value']
  if tag == 'Path':
    return child.attrib.get('value', '')
  if tag == 'PropertyPlacement':
    a = child.attrib
    return Placement(base=(float(a['Px']), float(a['Py']), float(a['Pz'])),
                     quat=(float(a['Q0']), float(a['Q1']), float(a['Q2']), float(a['Q3'])))
  if tag == 'PropertyVector':
    a = child.attrib
    return np.array([float(a['valueX']), float(a['valueY']), float(a['valueZ'])])
  if tag == 'Link':
    return child.attrib.get('value') or None
  if tag == 'LinkList':
    return [l.attrib['value'] for l in child.findall('Link')]
  if tag == 'LinkSubList':
    # [(object name, [sub-element names])], consecutive entries of one object merged
    # (App::PropertyLinkSubList.getValue groups them the same way)
    out = []
    for l in child.findall('Link'):
      name, sub = l.attrib.get('obj'), l.attrib.get('sub', '')
      if out and out[-1][0] == name:
        out[-1][1].append(sub)
      else:
        out.append((name, [sub]))
    return out
  if tag == 'XLink':
    # App::PropertyXLink: object `name` of this document, or of the document `file` (relative path)
    name, fname = child.attrib.get('name') or None, child.attrib.get('file') or ''
    return ExternalRef(fname, name) if (name and fname) else name
  if tag == 'LinkSub':
    return child.attrib.get('value') or None
  if tag == 'Python':
    out = dict(module=child.attrib.get('module'), **{'class': child.attrib.get('class')})
    raw = child.attrib.get('value', '')
    try:
      if child.attrib.get('encoded') == 'yes':
        raw = base64.b64decode(raw).decode()
      out['state'] = json.loads(raw) if raw else {}
    except Exception:
      out['state'] = {}
    return out
  if tag == 'PlacementList':
    fname = child.attrib.get('file')
    if not fname or zf is None or fname not in zf.namelist():
      return []
    data = zf.read(fname)
    (count,) = struct.unpack_from('<I', data, 0)
    vals = np.frombuffer(data, dtype='<f8', count=count * 7, offset=4).reshape(count, 7)
    return [Placement(base=v[:3], quat=v[3:]) for v in vals]
  if tag == 'BoolList':
    return [c == '1' for c in child.attrib.get('value', '')]
  if tag == 'Part':
    # Part::PropertyPartShape: the BRep text travels as a member of the zip; kept as bytes and
    # parsed only for objects that have no parametric recipe (scene/brep.py)
    fname = child.attrib.get('file')
    if not fname or zf is None or fname not in zf.namelist():
      return None
    data = zf.read(fname)
    return BRepPayload(fname, data) if data else None
  return None


class ExternalRef:
  """target of an App::PropertyXLink that lives in another FCStd file"""

  def __init__(self, file, name):
    self.file, self.name = file, name

  def __repr__(self):
    return f'<ExternalRef {self.file}#{self.name}>'


class BRepPayload:
  """the `<Object>.Shape.brp` member of an FCStd file (OpenCASCADE BRep text)"""

  def __init__(self, name, data):
    self.name, self.data = name, data

  def __repr__(self):
    return f'<BRepPayload {self.name}, {len(self.data)} bytes>'


# property tables of the workbench proxies (optical_group.py:29-96,
# point_source.py:32-70 + generic_source.py:23-37, simulation_settings.py:20-77)
_PROXY_PROPERTIES = {
  ('freecad.optics_design_workbench.freecad_elements.optical_group', 'OpticalGroupProxy'): (
      'OpticalType RefractiveIndex ReflectedProbabilityDensity RefractedProbabilityDensity PowerThetaDomain '
      'PowerPhiDomain RayModificationProbabilityDensity ModifyThetaDomain ModifyPhiDomain Reflectivity '
      'AbsorptionLength GratingType GratingLinesPerMillimeter GratingLinesOrientation GratingDiffractionOrder '
      'RecordHits').split(),
  ('freecad.optics_design_workbench.freecad_elements.point_source', 'PointSourceProxy'): (
      'PowerDensity Wavelength FocalLength Divergence ThetaDomain PhiDomain RadiusDomain RandomNumberGeneratorMode '
      'ThetaResolutionNumericMode RadiusResolutionNumericMode PhiResolutionNumericMode Fans FanPhi0 RaysPerFan '
      'FanModePowerSpan RecordRays IgnoredOpticalElements RaysPerIterationScale MaxIntersectionsScale '
      'MaxRayLengthScale').split(),
  ('freecad.optics_design_workbench.freecad_elements.surface_source', 'SurfaceSourceProxy'): (
      'ActiveSurfaces PowerDensity UVSamplingInitialResolution UVSamplingMaxRelAreaElementChange FanModeRayCount '
      'Wavelength ThetaDomain RandomNumberGeneratorMode ThetaResolutionNumericMode RadiusResolutionNumericMode '
      'FanModePowerSpan RecordRays IgnoredOpticalElements RaysPerIterationScale MaxIntersectionsScale '
      'MaxRayLengthScale').split(),
  ('freecad.optics_design_workbench.freecad_elements.replay_source', 'ReplaySourceProxy'): (
      'ReplayFromDir RecordRays IgnoredOpticalElements RaysPerIterationScale MaxIntersectionsScale '
      'MaxRayLengthScale Wavelength').split(),
  ('freecad.optics_design_workbench.freecad_elements.simulation_settings', 'SimulationSettingsProxy'): (
      'Active EnableStoreSingleShotData EndAfterIterations EndAfterRays EndAfterHits RaysPerIteration '
      'MaxIntersections DistanceTolerance MaxRayLength ShowRaysInContinuousMode WorkerProcessCount SequentialMode '
      'SequentialModeElements_00 StoreHitInitPoint StoreHitInitDirection StoreHitInitPower StoreHitInitWavelength '
      'StoreHitInitPhi StoreHitInitTheta StoreHitRayIndex StoreHitFanIndex StoreHitTotalFanCount '
      'StoreHitTotalRaysInFan').split(),
}


def repairProxies(doc):
  """documents saved without proxy information (`Proxy` = null) still carry
  the workbench properties; like the reference's repairAllProxies
  (freecad_elements/common.py:181-242) an object named Optical* whose
  properties match a proxy's table well enough (more than 5 present, fewer
  than 3 missing) gets that proxy back"""
  for obj in doc.Objects:
    if not obj.Name.startswith('Optical') or obj.ProxyClass:
      continue
    for (module, cls), names in _PROXY_PROPERTIES.items():
      present = sum(1 for n in names if n in obj._props)
      if present > 5 and present > len(names) - 3:
        obj._props['Proxy'] = {'module': module, 'class': cls, 'state': {}, 'repaired': True}
        break


class Document:
  """The parsed document: `doc.Objects`, `doc.getObject(name)`, `doc.<Name>`."""

  def __init__(self, path=None, _loaded=None):
    self.FileName = path
    self.Objects = []
    self._by_name = {}
    self._revision = 0
    # documents reached through links into other files (App::Link with an XLink target), by
    # absolute path; shared by all documents of one project so that a file is loaded once
    self._loaded = _loaded if _loaded is not None else {}
    self._external = []
    if path is not None:
      import os
      self._loaded[os.path.abspath(path)] = self
      self._load(path)

  def externalDocument(self, relpath):
    """the document a link points into (`relpath` relative to this file), or None if it is missing"""
    import os
    if self.FileName is None:
      return None
    full = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(self.FileName)), relpath))
    if full not in self._loaded:
      if not os.path.exists(full):
        return None
      Document(full, _loaded=self._loaded)
    ext = self._loaded[full]
    if ext is not self and ext not in self._external:
      self._external.append(ext)
    return ext

  def allDocuments(self):
    """this document and every document reachable from it through links
    (find._allObjects, freecad_elements/find.py:24-56)"""
    out, todo = [], [self]
    while todo:
      d = todo.pop(0)
      if d not in out:
        out.append(d)
        todo.extend(d._external)
    return out

  def allObjects(self):
    return [o for d in self.allDocuments() for o in d.Objects]

  # -- loading ------------------------------------------------------------
  def _load(self, path):
    with zipfile.ZipFile(path) as zf:
      root = ET.fromstring(zf.read('Document.xml'))
      self.ProgramVersion = root.attrib.get('ProgramVersion', '')
      for o in root.find('Objects').findall('Object'):
        obj = DocumentObject(self, o.attrib['name'], o.attrib['type'])
        self.Objects.append(obj)
        self._by_name[obj.Name] = obj
      for o in root.find('ObjectData').findall('Object'):
        obj = self._by_name.get(o.attrib['name'])
        props = o.find('Properties')
        if obj is None or props is None:
          continue
        for p in props.findall('Property'):
          try:
            val = _parse_property(p, zf)
          except Exception:
            val = None
          obj._props[p.attrib['name']] = val
          obj._types[p.attrib['name']] = p.attrib['type']
    # resolve link names to objects
    for obj in self.Objects:
      for k, v in list(obj._props.items()):
        t = obj._types.get(k, '')
        if isinstance(v, ExternalRef):
          ext = self.externalDocument(v.file)
          obj._props[k] = ext.getObject(v.name) if ext is not None else None
          if obj._props[k] is None:
            obj._props['_unresolved_' + k] = repr(v)
        elif t in ('App::PropertyLink', 'App::PropertyXLink', 'App::PropertyLinkGlobal'):
          obj._props[k] = self._by_name.get(v) if isinstance(v, str) else None
        elif t == 'App::PropertyLinkSubList':
          obj._props[k] = [(self._by_name[n], [x for x in subs if x]) for n, subs in (v or [])
                           if n in self._by_name]
        elif t in ('App::PropertyLinkList', 'App::PropertyLinkListGlobal'):
          obj._props[k] = [self._by_name[n] for n in (v or []) if n in self._by_name]
    repairProxies(self)
    self._revision = 0

  # -- API ----------------------------------------------------------------
  def _touch(self):
    self._revision += 1

  @property
  def revision(self):
    """changes when any document of the project changes"""
    return sum(d._revision for d in self.allDocuments())

  @property
  def shapesAsSaved(self):
    """no shape-defining property was written since the project was loaded"""
    return not any(getattr(d, '_shape_revision', 0) for d in self.allDocuments())

  def getObject(self, name):
    return self._by_name.get(name)

  def getObjectsByLabel(self, label):
    return [o for o in self.Objects if o._props.get('Label') == label]

  def addObject(self, type_id, name, **props):
    base, i = name, 0
    while name in self._by_name:
      i += 1
      name = f'{base}{i:03d}'
    obj = DocumentObject(self, name, type_id)
    obj._props.update(dict(Label=name, Placement=Placement.identity()))
    obj._props.update(props)
    self.Objects.append(obj)
    self._by_name[name] = obj
    self._touch()
    return obj

  def __getattr__(self, key):
    by_name = self.__dict__.get('_by_name', {})
    if key in by_name:
      return by_name[key]
    for o in self.__dict__.get('Objects', []):
      if o._props.get('Label') == key:
        return o
    raise AttributeError(key)

  def parents_of(self, obj):
    """objects that hold `obj` as a child (Group / ElementList containers)"""
    res = []
    for o in self.Objects:
      for key in ('Group', 'ElementList'):
        lst = o._props.get(key)
        if isinstance(lst, list) and any(c is obj for c in lst):
          res.append(o)
          break
    return res


def open_fcstd(path):
  return Document(path)
