"""Deterministic point placement following a tabulated 1-D density.

Mirror of the reference's helpers of the same names
(distributions/points_by_density.py:14-38), used by fan-mode ray placement
(`ScalarRandomVariable.findGrid`) and by the notebooks that check it:
  generatePointsWithGivenDensity1D  cell-centred cumulative density ->
                                    equidistant quantiles -> positions
  calcDiffDensity / calcHistDensity density estimates from point spacing /
                                    from a histogram
"""
import numpy as np


def calcHistDensity(X, bins=None):
  H, edges = np.histogram(X, **({} if bins is None else {'bins': bins}))
  return (edges[1:] + edges[:-1]) / 2, H / np.sum(H)


def calcDiffDensity(X):
  X = np.sort(np.asarray(X, dtype=np.float64))
  gaps = np.maximum(X[1:] - X[:-1], 1e-30)
  dens = 1 / gaps
  return (X[1:] + X[:-1]) / 2, dens / np.sum(dens)


def generatePointsWithGivenDensity1D(density, N, startFrom=None):
  """N points between X[0] and X[-1] whose local spacing follows 1/Y: the
  cumulative sum of Y lives on the cell boundaries around the samples, its
  equidistant quantiles are mapped back to positions; the two end points are
  the domain limits themselves"""
  X, Y = (np.asarray(a, dtype=np.float64) for a in density)
  nodes = np.empty(len(X) + 1)
  nodes[0] = X[0] - (X[1] - X[0]) / 2
  nodes[1:-1] = (X[:-1] + X[1:]) / 2
  nodes[-1] = X[-1] + (X[-1] - X[-2]) / 2
  cum = np.concatenate([[0], np.cumsum(Y)])
  cum = (cum - cum.min()) / (cum.max() - cum.min())
  quantiles = np.linspace(0, 1, int(round(N)))[1:-1]
  return np.concatenate([[X[0]], np.interp(quantiles, cum, nodes), [X[-1]]])
