"""2-D cartesian / polar hit histograms on the host.

Same constructor arguments, attributes (`hist`, `binX`, `binY`, `X`, `Y`,
`binAreas`) and `byAzimuth()` as the reference's `Histogram`
(jupyter_utils/histogram.py:19-89, 150-161), with its `plot` / `plotByAzimuth` on explicit matplotlib axes (:91-148, 163-166).
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

  @classmethod
  def fromBinned(cls, hist, binX, binY, planeNormal, xInPlaneVec, origin, binCoords):
    """a Histogram around counts that were binned elsewhere (on the device, `DeviceHits.histogram`)
    with this class' rules: same attributes as after __init__"""
    self = cls.__new__(cls)
    self._planeNormal, self._xInPlaneVec, self._origin = planeNormal, xInPlaneVec, np.asarray(origin)
    self._binCoords = binCoords
    self.hist, self.binX, self.binY = np.asarray(hist, dtype=np.float64), np.asarray(binX), np.asarray(binY)
    if binCoords == 'cartesian':
      self.X = (self.binX[1:] + self.binX[:-1]) / 2
      self.Y = (self.binY[1:] + self.binY[:-1]) / 2
      self.binAreas = 1
    else:
      dphi = np.diff(self.binX)[:, None]
      r1, r2 = self.binY[:-1][None, :], self.binY[1:][None, :]
      self.binAreas = dphi * (r1 + r2) / 2 * (r2 - r1)
    return self

  def byAzimuth(self):
    '''
    Return histograms for each azimuthal angle bin. Only available in polar mode.
    '''
    if self._binCoords != 'polar':
      raise ValueError('byAzimuth is only available for polar histograms '
                       '(created with binCoords="polar" argument)')
    return ((self.binX[1:] + self.binX[:-1]) / 2, (self.binY[:-1] + self.binY[1:]) / 2,
            self.hist / self.binAreas)

  def scaledHist(self, scale='max'):
    """hit density per bin area, transposed for plotting; scale 'max' normalises to the peak,
    a number divides by it, None leaves the density as it is (Histogram.plot, histogram.py:108-113)"""
    h = (self.hist / self.binAreas).T
    if scale == 'max':
      return h / h.max()
    return h if scale is None else h / scale

  def plot(self, cbar={}, title=None, scale='max', **kwargs):
    """pcolormesh of the density on the current matplotlib axes (polar axes for polar bins;
    histogram.py:91-148); cbar: keyword arguments of the colour bar, anything but a dict: none"""
    import matplotlib.pyplot as plt
    polar = self._binCoords == 'polar'
    ax = plt.gca()
    if (ax.name == 'polar') != polar:
      fig = ax.figure
      spec = ax.get_subplotspec()
      ax.remove()
      ax = fig.add_subplot(spec, projection='polar' if polar else 'rectilinear')
      plt.sca(ax)
    h = self.scaledHist(scale)
    bx = self.binX
    if polar:
      up = int(np.ceil(200 / len(bx)))          # finer azimuth steps: round cells
      if up > 1:
        bx = np.concatenate([np.linspace(a, b, up + 1)[:-1] for a, b in zip(bx[:-1], bx[1:])] + [bx[-1:]])
        h = np.repeat(h, up, axis=1)
    mesh = ax.pcolormesh(*np.meshgrid(bx, self.binY), h, **kwargs)
    if isinstance(cbar, dict):
      plt.colorbar(mesh, ax=ax, **cbar).set_label('hit density per bin')
    if title is None:
      n, p, o = self._planeNormal, self._xInPlaneVec, self._origin
      title = (f'plane normal = [{n[0]:.2f}, {n[1]:.2f}, {n[2]:.2f}],\n'
               f'projected $x$ = [{p[0]:.2f}, {p[1]:.2f}, {p[2]:.2f}]'
               + ('' if np.allclose(o, 0) else f',\norigin = [{o[0]:.2e}, {o[1]:.2e}]'))
    ax.set_title(title, fontsize=10, **(dict(y=1.09) if polar else {}))
    if not polar:
      ax.axis('equal')
      ax.set_xlabel(r'projected $x$')
      ax.set_ylabel(r'projected $y$')
    ax.set_aspect('equal')
    return mesh

  def plotByAzimuth(self):
    """radial density profiles, one line per azimuth bin (histogram.py:163-166)"""
    import matplotlib.pyplot as plt
    phi, r, h = self.byAzimuth()
    for p, row in zip(phi, h):
      plt.plot(r, row, label=f'$\\phi={p / np.pi:.1f}\\pi$')
    plt.xlabel('radius $r$')
    plt.ylabel('hit density per bin')
    plt.legend()
