"""2-D cartesian / polar hit histograms on the host.

Same constructor arguments, attributes (`hist`, `binX`, `binY`, `X`, `Y`,
`binAreas`) and `byAzimuth()` as the reference's `Histogram`
(jupyter_utils/histogram.py:19-89, 150-161); plotting is left out.
"""
import numpy as np


def _radius_bins(kwargs, radius, polar):
  bins = kwargs.pop('bins', 50)
  if hasattr(bins, '__len__') and len(bins) > 2:
    bins = len(bins)
  bins = list(bins) if hasattr(bins, '__len__') else [bins, bins]
  if polar:
    bins[1] = np.linspace(0, radius, bins[1])
  else:
    bins = [np.linspace(-radius, radius, bins[0]), np.linspace(-radius, radius, bins[1])]
  kwargs['bins'] = bins


class Histogram:
  '''
  Class representing a 2D polar or cartesian histogram.
  '''

  def __init__(self, X, Y, planeNormal, xInPlaneVec, radius=None, binCoords='cartesian',
               origin=None, **kwargs):
    self._planeNormal = planeNormal
    self._xInPlaneVec = xInPlaneVec
    self._origin = np.array([np.median(X), np.median(Y)]) if origin is None else origin
    X = X - self._origin[0]
    Y = Y - self._origin[1]
    mode = binCoords.lower()
    if mode in 'cartesian':
      self._binCoords = 'cartesian'
      if radius is not None:
        _radius_bins(kwargs, radius, polar=False)
      self.hist, self.binX, self.binY = np.histogram2d(X, Y, **kwargs)
      self.X = (self.binX[1:] + self.binX[:-1]) / 2
      self.Y = (self.binY[1:] + self.binY[:-1]) / 2
      self.binAreas = 1
    elif mode in 'polar':
      self._binCoords = 'polar'
      if radius is not None:
        _radius_bins(kwargs, radius, polar=True)
      self.hist, self.binX, self.binY = np.histogram2d(np.arctan2(X, Y), np.sqrt(X**2 + Y**2), **kwargs)
      dphi = np.diff(self.binX)[:, None]
      r1, r2 = self.binY[:-1][None, :], self.binY[1:][None, :]
      self.binAreas = dphi * (r1 + r2) / 2 * (r2 - r1)
    else:
      raise ValueError(f'found invalid binCoord mode {binCoords!r}, expect one of "cartesian" or "polar"')

  def byAzimuth(self):
    '''
    Return histograms for each azimuthal angle bin. Only available in polar mode.
    '''
    if self._binCoords != 'polar':
      raise ValueError('byAzimuth is only available for polar histograms '
                       '(created with binCoords="polar" argument)')
    return ((self.binX[1:] + self.binX[:-1]) / 2, (self.binY[:-1] + self.binY[1:]) / 2,
            self.hist / self.binAreas)
