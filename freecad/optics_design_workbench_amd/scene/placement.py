"""Rigid placements as 4x4 float64 matrices (FreeCAD `Placement` semantics).

FreeCAD stores a placement as a translation (Px,Py,Pz) and a unit quaternion
(Q0,Q1,Q2,Q3) = (x,y,z,w); the reference turns them into matrices with
`placement.toMatrix()` (freecad_elements/common.py:112-125).
"""
import numpy as np


def quaternion_matrix(q0, q1, q2, q3):
  x, y, z, w = float(q0), float(q1), float(q2), float(q3)
  n = np.sqrt(x * x + y * y + z * z + w * w)
  if n == 0:
    return np.eye(3)
  x, y, z, w = x / n, y / n, z / n, w / n
  return np.array([
      [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def from_axis_angle(axis, angle_rad):
  a = np.asarray(axis, dtype=np.float64)
  a = a / np.linalg.norm(a)
  s, c = np.sin(angle_rad / 2), np.cos(angle_rad / 2)
  return quaternion_matrix(a[0] * s, a[1] * s, a[2] * s, c)


class Placement:
  """Immutable rigid transform; `a * b` applies b first, like FreeCAD."""
  __slots__ = ('m',)

  def __init__(self, base=(0, 0, 0), quat=(0, 0, 0, 1), matrix=None):
    if matrix is not None:
      self.m = np.array(matrix, dtype=np.float64).reshape(4, 4)
    else:
      self.m = np.eye(4)
      self.m[:3, :3] = quaternion_matrix(*quat)
      self.m[:3, 3] = base

  @classmethod
  def identity(cls):
    return cls()

  def __mul__(self, other):
    if isinstance(other, Placement):
      return Placement(matrix=self.m @ other.m)
    v = np.asarray(other, dtype=np.float64)
    return self.m[:3, :3] @ v + self.m[:3, 3]

  def inverse(self):
    inv = np.eye(4)
    rt = self.m[:3, :3].T
    inv[:3, :3] = rt
    inv[:3, 3] = -rt @ self.m[:3, 3]
    return Placement(matrix=inv)

  def toMatrix(self):
    return self.m.copy()

  @property
  def Base(self):
    return self.m[:3, 3].copy()

  @property
  def Rotation(self):
    return self.m[:3, :3].copy()

  def axisAngle(self):
    """(unit axis, angle in radians) with the angle in [0, pi] like FreeCAD's
    Rotation.Axis / Rotation.Angle; the identity reports the z axis"""
    r = self.m[:3, :3]
    angle = float(np.arccos(np.clip((np.trace(r) - 1) / 2, -1, 1)))
    v = np.array([r[2, 1] - r[1, 2], r[0, 2] - r[2, 0], r[1, 0] - r[0, 1]])
    n = np.linalg.norm(v)
    if n > 1e-12:
      return v / n, angle
    if angle < 1e-6:
      return np.array([0.0, 0.0, 1.0]), 0.0
    # angle = pi: axis from the symmetric part
    d = np.clip((np.diag(r) + 1) / 2, 0, None)
    axis = np.sqrt(d)
    k = int(np.argmax(axis))
    for j in range(3):
      if j != k and r[k, j] + r[j, k] < 0:
        axis[j] = -axis[j]
    return axis / np.linalg.norm(axis), angle

  def withRotation(self, axis, angle_rad):
    """same Base, rotation about `axis` by `angle_rad` (any real number, like
    assigning FreeCAD's Rotation.Angle)"""
    out = Placement(matrix=self.m)
    out.m[:3, :3] = from_axis_angle(axis, angle_rad)
    return out

  def withBase(self, base):
    out = Placement(matrix=self.m)
    out.m[:3, 3] = np.asarray(base, dtype=np.float64)
    return out

  def rows12(self):
    """(R|t) rows, the 12-double layout of odw_trace.h"""
    return np.ascontiguousarray(self.m[:3, :]).reshape(12)

  def __repr__(self):
    return f'Placement(base={self.Base.tolist()}, R={self.Rotation.tolist()})'
