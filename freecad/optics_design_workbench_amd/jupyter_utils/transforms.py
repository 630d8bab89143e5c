"""4x4 transforms on (n, 3) point arrays (jupyter_utils/transforms.py): the matrices of
`global-info.pkl` (allCoordinateTransformMatrices) applied to hit coordinates"""
import numpy as np


def applyTransformation(points, transform):
  t = np.asarray(transform, dtype=np.float64)
  return np.asarray(points, dtype=np.float64) @ t[:3, :3].T + t[:3, 3]


def applyTransformationWithoutTranslation(points, transform):
  return np.asarray(points, dtype=np.float64) @ np.asarray(transform, dtype=np.float64)[:3, :3].T
