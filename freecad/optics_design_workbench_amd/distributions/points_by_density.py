"""Deterministic point placement following a tabulated 1-D density.

Counterparts of the reference's helpers of the same names
(distributions/points_by_density.py:14-38), used by fan-mode ray placement
(`ScalarRandomVariable.findGrid`) and by the notebooks that check it:
  generatePointsWithGivenDensity1D  cell-centred cumulative density ->
                                    equidistant quantiles -> positions
  calcDiffDensity / calcHistDensity density estimates from point spacing /
                                    from a histogram
The arithmetic (order of operations included) is the reference's: the fan
goldens compare bit for bit.
"""
import numpy as np


def _between(a):
  """mid-points of neighbouring entries"""
  return (a[1:] + a[:-1]) / 2


def _as_weights(w):
  return w / np.sum(w)


def _cells(x):
  """boundaries of the cells around the samples x: half a step beyond both ends"""
  edges = np.empty(len(x) + 1)
  edges[1:-1] = _between(x)
  edges[0] = x[0] - (x[1] - x[0]) / 2
  edges[-1] = x[-1] + (x[-1] - x[-2]) / 2
  return edges


def calcHistDensity(X, bins=None):
  options = {} if bins is None else dict(bins=bins)
  counts, edges = np.histogram(X, **options)
  return _between(edges), _as_weights(counts)


def calcDiffDensity(X):
  ordered = np.sort(np.asarray(X, dtype=np.float64))
  spacing = np.maximum(np.diff(ordered), 1e-30)
  return _between(ordered), _as_weights(1 / spacing)


def generatePointsWithGivenDensity1D(density, N, startFrom=None):
  """N points from X[0] to X[-1] spaced like 1/Y.  The running sum of Y belongs to the cell
  boundaries around the samples; brought to [0, 1] it is inverted at equidistant levels.  The
  outermost two levels are replaced by the domain limits themselves."""
  X = np.asarray(density[0], dtype=np.float64)
  running = np.zeros(len(X) + 1)
  running[1:] = np.cumsum(np.asarray(density[1], dtype=np.float64))
  low, high = running.min(), running.max()
  running = (running - low) / (high - low)
  inner = np.interp(np.linspace(0, 1, int(round(N)))[1:-1], running, _cells(X))
  out = np.empty(len(inner) + 2)
  out[0], out[-1] = X[0], X[-1]
  out[1:-1] = inner
  return out
