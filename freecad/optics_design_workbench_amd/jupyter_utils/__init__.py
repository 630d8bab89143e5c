"""notebook-facing API mirroring the reference's jupyter_utils
(jupyter_utils/__init__.py:11-16): FreecadDocument, RawFolder, Hits, Histogram"""
from .hits import Hits
from .histogram import Histogram
from .transforms import applyTransformation, applyTransformationWithoutTranslation


def __getattr__(name):
  # FreecadDocument pulls in the simulation package (and with it the native
  # binding); keep `from ...jupyter_utils import Hits` free of that import
  if name in ('FreecadDocument', 'FreecadObject', 'FreecadProperty'):
    from . import freecad_document
    return getattr(freecad_document, name)
  if name in ('RawFolder', 'RawFolderRange', 'rawFolders', 'rawFolderByIndex', 'latestRawFolder'):
    from ..simulation import results_store
    return getattr(results_store, name)
  raise AttributeError(name)
