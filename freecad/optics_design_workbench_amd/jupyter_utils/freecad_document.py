"""`FreecadDocument`: the notebook-facing document handle.

In the reference this class drives a `FreeCAD -c` child process over pipes
(jupyter_utils/freecad_document.py:454-1288) and `runSimulation` waits for the
run folder that process writes (:640-770).  Here the document is parsed
in-process (FCStd-lite) and the simulation runs on the GPU tracer; the
surface a notebook touches is kept:

    with FreecadDocument('GettingStarted.FCStd') as f:
      f.Sphere.Radius = 10.5                      # property write
      print(f.OpticalPointSource.PowerDensity.get())
      raw = f.runSimulation('true')               # -> RawFolder
      hits = raw.loadHits('*')                    # -> Hits
      hits.histogram(bins=30)

Objects resolve by label first, then by internal name, as in the reference
(`getObjectsByLabel(...)[0]`, freecad_document.py:197-200).
"""
import os
import shutil
import tempfile

import numpy as np

from ..scene.fcstd import Document
from ..simulation import results_store, simulation_loop


class _RotationView:
  """what `Placement.Rotation` looks like from a notebook: Axis, Angle (radians)"""

  def __init__(self, placement):
    self.Axis, self.Angle = placement.axisAngle()

  def __repr__(self):
    return f'Rotation (axis={self.Axis.tolist()}, angle={self.Angle})'


class FreecadProperty:
  '''
  One property of a document object; `get()`/`set()` like the reference's
  FreecadProperty (freecad_document.py:176-413).  Comparisons, float() and
  str() act on the value.  Attribute paths below a Placement can be read and
  assigned like in a FreeCAD shell (`f.Source.Placement.Rotation.Angle = 0.3`,
  `f.Box.Placement.Base.z = 12`): the reference forwards such lines to FreeCAD
  (:225-232), here the placement is rebuilt and written back.
  '''

  def __init__(self, obj, name, path=()):
    self.__dict__['_obj'] = obj
    self.__dict__['_name'] = name
    self.__dict__['_path'] = tuple(path)

  def get(self):
    from ..scene.placement import Placement
    v = self._obj._props[self._name]
    for key in self._path:
      if isinstance(v, Placement) and key == 'Rotation':
        v = _RotationView(v)
      elif isinstance(v, np.ndarray) and key in ('x', 'y', 'z'):
        v = float(v['xyz'.index(key)])
      else:
        v = getattr(v, key)
    return v

  def set(self, value):
    from ..scene.placement import Placement
    if isinstance(value, FreecadProperty):
      value = value.get()
    if not self._path:
      setattr(self._obj, self._name, value)
      return
    root = self._obj._props[self._name]
    path = self._path
    if isinstance(root, Placement):
      axis, angle = root.axisAngle()
      if path == ('Base',):
        new = root.withBase(value)
      elif len(path) == 2 and path[0] == 'Base' and path[1] in ('x', 'y', 'z'):
        b = root.Base
        b['xyz'.index(path[1])] = float(value)
        new = root.withBase(b)
      elif path == ('Rotation', 'Angle'):
        new = root.withRotation(axis, float(value))        # radians, as in FreeCAD
      elif path == ('Rotation', 'Axis'):
        new = root.withRotation(value, angle)
      else:
        raise AttributeError(f'cannot assign {self._obj.Name}.{self._name}.{".".join(path)}')
      setattr(self._obj, self._name, new)
      return
    if isinstance(root, np.ndarray) and len(path) == 1 and path[0] in ('x', 'y', 'z'):
      new = root.copy()
      new['xyz'.index(path[0])] = float(value)
      setattr(self._obj, self._name, new)
      return
    raise AttributeError(f'cannot assign {self._obj.Name}.{self._name}.{".".join(path)}')

  def getStr(self):
    return str(self.get())

  def getFloat(self):
    return float(self.get())

  def getInt(self):
    return int(self.get())

  def __repr__(self):
    return f'<FreecadProperty {self._obj.Name}.{".".join((self._name,) + self._path)}, value: {self.getStr()}>'

  def __float__(self):
    return float(self.get())

  def __eq__(self, other):
    return self.get() == (other.get() if isinstance(other, FreecadProperty) else other)

  def __getattr__(self, key):            # e.g. Placement.Base, Placement.Rotation.Angle
    if key.startswith('__'):
      raise AttributeError(key)
    value = self.get()
    from ..scene.placement import Placement
    if isinstance(value, (Placement, _RotationView)) or (isinstance(value, np.ndarray) and key in ('x', 'y', 'z')):
      return FreecadProperty(self._obj, self._name, self._path + (key,))
    return getattr(value, key)

  def __setattr__(self, key, value):
    FreecadProperty(self._obj, self._name, self._path + (key,)).set(value)


class FreecadObject:
  '''
  A document object; attribute reads give FreecadProperty handles, attribute
  writes set the property (and mark the scene for re-baking).
  '''

  def __init__(self, obj):
    self.__dict__['_obj'] = obj

  def __getattr__(self, key):
    if key not in self._obj._props:
      raise AttributeError(f'{self._obj.Name} has no property {key!r}')
    return FreecadProperty(self._obj, key)

  def __setattr__(self, key, value):
    setattr(self._obj, key, value.get() if isinstance(value, FreecadProperty) else value)

  def __dir__(self):
    return self._obj.PropertiesList

  def __repr__(self):
    return f'<FreecadObject {self._obj.Name} ({self._obj._props.get("Label", "")})>'


class FreecadDocument:
  '''
  Handle on one FCStd project.  workInTempCopy=True (the reference's option of
  the same name) leaves the original project folder untouched: results go to a
  temporary `.OpticsDesign` folder that is removed on close.
  '''

  def __init__(self, path, workInTempCopy=False, device=0, seed=simulation_loop.DEFAULT_SEED):
    self._origPath = os.path.abspath(path)
    self._tmp = None
    if workInTempCopy:
      self._tmp = tempfile.mkdtemp(prefix='odw-')
      work = os.path.join(self._tmp, os.path.basename(path))
      shutil.copy(self._origPath, work)
      self._path = work
    else:
      self._path = self._origPath
    self.__dict__['_doc'] = Document(self._path)
    self._resultsPath = results_store.resultsFolderPath(self._path)
    self._device = device
    self._seed = seed
    self._runs = 0

  # -- context manager / lifecycle -------------------------------------------
  def __enter__(self):
    return self

  def __exit__(self, *a):
    self.close()

  def close(self):
    if self._tmp and os.path.isdir(self._tmp):
      shutil.rmtree(self._tmp, ignore_errors=True)
    self._tmp = None

  def path(self):
    return self._path

  def resultsPath(self):
    return self._resultsPath

  def isWorkInTempCopy(self):
    return self._tmp is not None

  def purgeTempFolder(self):
    """remove the temporary working copy's results (freecad_document.py:546-552)"""
    if self._tmp and os.path.isdir(self._resultsPath):
      shutil.rmtree(self._resultsPath, ignore_errors=True)

  def disableFastMode(self):
    """the reference's fast mode skips document recomputes in its FreeCAD child process
    (freecad_document.py:940-972); there is no such process here: nothing to switch"""

  def open(self):
    return self

  def isRunning(self):
    return True

  def save(self):
    raise NotImplementedError('writing FCStd files needs FreeCAD; property changes live in this session only')

  def __repr__(self):
    return f'<FreecadDocument {os.path.basename(self._path)}>'

  # -- objects -----------------------------------------------------------------
  def getObject(self, nameOrLabel):
    doc = self.__dict__['_doc']
    by_label = doc.getObjectsByLabel(nameOrLabel)
    obj = by_label[0] if by_label else doc.getObject(nameOrLabel)
    if obj is None:
      raise AttributeError(f'document has no object labelled or named {nameOrLabel!r}')
    return FreecadObject(obj)

  def __getattr__(self, key):
    if key.startswith('_'):
      raise AttributeError(key)
    return self.getObject(key)

  def objects(self):
    return [FreecadObject(o) for o in self.__dict__['_doc'].Objects]

  def document(self):
    return self.__dict__['_doc']

  # -- simulation ----------------------------------------------------------------
  def runSimulation(self, action='true', endIf=None, **kwargs):
    """run and return the RawFolder of the new run
    (freecad_document.py:640-770); endIf(rawFolder) -> bool ends a continuous
    run early"""
    allowed = 'true pseudo singletrue singlepseudo fans'.split() + ['singlefans']
    if action not in allowed:
      raise ValueError(f'illegal action {action}, expected one of {", ".join(allowed)}')
    store_box = {}

    def _end_if(store):
      store.flush()
      return bool(endIf(results_store.RawFolder(store.runFolderPath())))

    # every run draws from a fresh part of the Philox stream
    seed = self._seed + self._runs
    self._runs += 1
    store = simulation_loop.runSimulation(self.__dict__['_doc'], action, seed=seed, device=self._device,
                                          resultsPath=self._resultsPath,
                                          endIf=_end_if if endIf is not None else None, **kwargs)
    store_box['store'] = store
    return results_store.RawFolder(store.runFolderPath())

  def rawFolders(self):
    return results_store.rawFolders(self._resultsPath)

  def latestRawFolder(self):
    return results_store.latestRawFolder(self._resultsPath)

  def rawFolderByIndex(self, index=-1):
    return results_store.rawFolderByIndex(index, self._resultsPath)
