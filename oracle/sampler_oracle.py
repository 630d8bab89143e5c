"""numpy restatement of the reference's numeric-mode sampler tables and draw.

TEST INFRASTRUCTURE ONLY (see oracle/odw_oracle.c header for the rule).

Follows, literally and without optimisation,
  distributions/random_number_generator.py:323-334  _numericalResolution (odd)
  distributions/random_number_generator.py:337-369  _generateNumericScalarLambda
  distributions/random_number_generator.py:372-464  _lambdasFromSampled
  distributions/random_number_generator.py:467-560  draw (uniform order)
  distributions/random_number_generator.py:685-725  findGrid
  distributions/points_by_density.py:25-38          generatePointsWithGivenDensity1D
Pinned bit-for-bit by tests/golden/sampler_*.npz and fan_grid.npz, which were
produced by the reference's own module (tests/golden/make_golden.py).
"""
import numpy as np
import sympy as sy


def odd_resolution(res):
  res = int(round(res))
  return res + 1 if res % 2 == 0 else res


def build_tables(density, order, domains, resolutions):
  """-> dict(edges=[per var], cdf=[per var]) exactly as the reference holds
  them after compile(): cdf[0] has shape (n1-1, n0) (conditional on var 1),
  cdf[1] has shape (n1,), both UNnormalised cumulative sums."""
  expr = sy.sympify(density)
  syms = []
  for name in order:
    l1, l2 = domains[name]
    kw = dict(nonnegative=True) if l1 >= 0 else dict(nonpositive=True) if l2 <= 0 else {}
    s = sy.Symbol(name, real=True, **kw)
    expr = expr.subs(sy.Symbol(name), s)
    syms.append(s)
  ranges, mids = [], []
  for name in order:
    l1, l2 = domains[name]
    r = np.linspace(l1, l2, odd_resolution(resolutions[name]))
    ranges.append(r)
    mids.append((r[1:] + r[:-1]) / 2)
  grids = np.meshgrid(*mids)
  lam = sy.lambdify(syms, expr, modules=['numpy', 'scipy'])
  probs = lam(*grids)
  if not hasattr(probs, 'shape') or np.shape(probs) != np.shape(grids[0]):
    probs = grids[0] * 0 + probs
  if (probs < 0).any():
    raise ValueError('negative probability density')
  cdfs = []
  for var_i in range(len(order)):
    g = probs
    for _ in range(var_i):
      g = g.sum(axis=-1)
    g = np.insert(g, 0, np.zeros(g.shape[:-1]), axis=-1)
    g = np.cumsum(g, axis=-1)
    cdfs.append(g)
  return dict(edges=ranges, mids=mids, cdf=cdfs)


def draw_from_uniforms(tables, u_last, u_first):
  """two-variable draw: the LAST variable (phi) is drawn first from u_last,
  then the first variable (theta/r) conditional on it from u_first."""
  edges, mids, cdf = tables['edges'], tables['mids'], tables['cdf']
  c1 = cdf[1] / cdf[1][-1]
  v1 = np.interp(u_last, c1, edges[1])
  v0 = np.empty_like(v1)
  for i in range(len(v1)):
    row = np.argmin(np.abs(mids[1] - v1[i]))
    col = cdf[0][row, :]
    col = col / col[-1]
    v0[i] = np.interp(u_first[i], col, edges[0])
  return v0, v1


def find_grid(density, var, domain, resolution, N):
  """ScalarRandomVariable.findGrid + generatePointsWithGivenDensity1D"""
  l1, l2 = domain
  s = sy.Symbol(var, real=True, **(dict(nonnegative=True) if l1 >= 0 else
                                   dict(nonpositive=True) if l2 <= 0 else {}))
  expr = sy.sympify(density).subs(sy.Symbol(var), s)
  X = np.linspace(l1, l2, odd_resolution(resolution))
  Y = sy.lambdify(s, expr, modules=['numpy', 'scipy'])(X)
  if not hasattr(Y, 'shape'):
    Y = Y * np.ones(X.shape)
  Xi = np.concatenate([[X[0] - (X[1] - X[0]) / 2], (X[:-1] + X[1:]) / 2,
                       [X[-1] + (X[-1] - X[-2]) / 2]])
  Yi = np.concatenate([[0], np.cumsum(Y)])
  Yi = (Yi - Yi.min()) / (Yi.max() - Yi.min())
  pick = np.linspace(0, 1, int(round(N)))[1:-1]
  res = np.concatenate([[X[0]], np.interp(pick, Yi, Xi), [X[-1]]])
  return res[np.logical_and(X.min() <= res, res <= X.max())]
