"""Coordinate changes for hit arrays.

`global-info.pkl` carries, for every optical object and light source, the 4x4
matrices between its local frame and the global one
(scene/bake.py:allCoordinateTransformMatrices).  These two helpers push (n, 3)
arrays through such a matrix: positions (rotation + shift) and directions
(rotation only).  Same names and argument order as the reference's
jupyter_utils.transforms, so notebooks keep working.
"""
import numpy as np


def _rows(points):
  p = np.asarray(points, dtype=np.float64)
  return p.reshape(-1, 3), p.shape


def applyTransformation(points, transform):
  """positions: homogeneous coordinates (x, y, z, 1) times the matrix"""
  p, shape = _rows(points)
  t = np.asarray(transform, dtype=np.float64)
  h = np.concatenate([p, np.ones((len(p), 1))], axis=1)
  return np.einsum('ij,nj->ni', t[:3, :], h).reshape(shape)


def applyTransformationWithoutTranslation(points, transform):
  """directions: homogeneous coordinates (x, y, z, 0): the shift drops out"""
  p, shape = _rows(points)
  t = np.asarray(transform, dtype=np.float64)
  h = np.concatenate([p, np.zeros((len(p), 1))], axis=1)
  return np.einsum('ij,nj->ni', t[:3, :], h).reshape(shape)
