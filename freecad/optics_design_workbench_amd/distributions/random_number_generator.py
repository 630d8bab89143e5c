"""Tabulated inverse-CDF random variables (host side of ray generation).

Mirror of the reference's `VectorRandomVariable` / `ScalarRandomVariable`
interface (distributions/random_number_generator.py:54-769) restricted to what
the device needs.  The reference first tries a sympy analytic inverse CDF for
2 s and falls back to numeric mode (random_number_generator.py:72-119); every
BASELINE source stores `RandomNumberGeneratorMode = numeric` (sympy does not
invert their densities).  Both modes are here (round 5): `compile()` makes the
same attempt (distributions/analytic.py), `mode()` reports its outcome like the
reference's, host draws use the closed form, and the device's table is then
built FROM the closed form -- cdf values exact at the numeric mode's edges --
instead of from mid-point sums.  `compile(disableAnalytical=True)` is numeric
mode as before, bit for bit.

Table definition (random_number_generator.py:337-369, 372-464), reproduced
bit for bit (tests/test_distributions.py against tests/golden/sampler_*.npz):
  edges_v = linspace(l1, l2, res_v)           res forced odd
  mid_v   = (edges_v[1:] + edges_v[:-1]) / 2
  P       = density(mid_theta x mid_phi)      shape (nphi-1, ntheta-1)
  cdf_theta[row] = [0, cumsum(P[row])] / total        one row per phi mid
  cdf_phi        = [0, cumsum(P.sum(axis=-1))] / total
and the draw: phi = interp(u_phi, cdf_phi, edges_phi); row = argmin|mid_phi -
phi|; theta = interp(u_theta, cdf_theta[row], edges_theta).
"""
import numpy as np
import sympy as sy


def _odd(res):
  res = int(round(float(res)))
  return res + 1 if res % 2 == 0 else res


class SamplerTables:
  """what is uploaded to the device (odw_source_desc tables)"""

  def __init__(self, first_edges, first_cdf, last_edges, last_cdf):
    self.t_edges = np.ascontiguousarray(first_edges, dtype=np.float64)
    self.t_cdf = np.ascontiguousarray(first_cdf, dtype=np.float64)      # (rows, knots)
    self.phi_edges = np.ascontiguousarray(last_edges, dtype=np.float64)
    self.phi_cdf = np.ascontiguousarray(last_cdf, dtype=np.float64)

  @property
  def n_rows(self):
    return self.t_cdf.shape[0]

  def draw(self, u_last, u_first):
    """host evaluation of the device sampler (same arithmetic)"""
    v1 = np.interp(u_last, self.phi_cdf, self.phi_edges)
    if self.n_rows == 1:
      return np.interp(u_first, self.t_cdf[0], self.t_edges), v1
    mids = (self.phi_edges[1:] + self.phi_edges[:-1]) / 2
    v1 = np.asarray(v1)
    # argmin_i |mid_i - v1| (first minimum), evaluated on the cell containing v1 and its two neighbours
    cell = np.clip(np.searchsorted(self.phi_edges, v1, side='right') - 1, 0, len(mids) - 1)
    cand = np.clip(cell[:, None] + np.array([-1, 0, 1])[None, :], 0, len(mids) - 1)
    rows = cand[np.arange(len(v1)), np.abs(mids[cand] - v1[:, None]).argmin(axis=1)]
    v0 = np.empty_like(v1)
    order = np.argsort(rows, kind='stable')
    bounds = np.searchsorted(rows[order], np.arange(self.n_rows + 1))
    for r in np.nonzero(bounds[1:] > bounds[:-1])[0]:
      idx = order[bounds[r]:bounds[r + 1]]
      v0[idx] = np.interp(u_first[idx], self.t_cdf[r], self.t_edges)
    return v0, v1


def _as_cdf(values):
  """a closed-form cdf evaluated on the edges, as a table column: finite, from 0 to 1, not decreasing"""
  c = np.asarray(values, dtype=np.float64)
  if not np.all(np.isfinite(c)) or abs(c[0]) > 1e-9 or abs(c[-1] - 1.0) > 1e-9:
    raise ValueError('closed-form cdf does not run from 0 to 1 over the domain')
  c = np.clip(c, 0.0, 1.0)
  c[0], c[-1] = 0.0, 1.0
  if np.any(np.diff(c) < -1e-12):
    raise ValueError('closed-form cdf decreases')
  return np.maximum.accumulate(c)


class VectorRandomVariable:
  '''
  Vector valued random variable (two variables).
  '''

  def __init__(self, probabilityDensity, variableDomains={}, numericalResolutions={},
               variableOrder=None, warnIfDiscretizationStepAbove=5e-2):
    self._probabilityDensity = probabilityDensity
    self._variableDomains = dict(variableDomains)
    self._numericalResolutions = numericalResolutions
    self._variableOrder = list(variableOrder) if variableOrder else None
    self._constantsDict = {}
    self._mode = 'not yet compiled'
    self._tables = None
    self._inverses = None
    self._analytic_off = False

  def mode(self):
    return self._mode

  def _resolution(self, name, nvars):
    res = self._numericalResolutions
    if not res:
      res = 5 + int(1e6**(1 / nvars))
    if isinstance(res, dict):
      res = res.get(name)
    return _odd(res)

  def _prepare(self, **constants):
    expr = sy.sympify(self._probabilityDensity)
    used = {}
    for name, val in constants.items():
      if name in [str(s) for s in expr.free_symbols]:
        expr = expr.subs(name, val)
        used[name] = val
    names = [str(s) for s in expr.free_symbols]
    order = [n for n in (self._variableOrder or []) if n in names]
    order += [n for n in names if n not in order]
    order += [n for n in self._variableDomains if n not in order]
    syms = []
    for name in order:
      l1, l2 = self._variableDomains.get(name, (-np.inf, np.inf))
      if not np.isfinite(l1) or not np.isfinite(l2):
        raise ValueError(f'numerical solution requires finite limits, but found limits '
                         f'[{l1}, {l2}] for variable {name}')
      kw = dict(nonnegative=True) if l1 >= 0 else dict(nonpositive=True) if l2 <= 0 else {}
      s = sy.Symbol(name, real=True, **kw)
      expr = expr.subs(sy.Symbol(name), s)
      syms.append(s)
    for s in expr.free_symbols:
      if s not in syms:
        raise ValueError(f'probabilty density expression {expr} has free symbol {s} '
                         f'which is not in list of variables {syms}')
    if expr.find(sy.DiracDelta):
      raise ValueError('cannot use numeric mode for expression containing DiracDelta')
    return expr, syms, order, used

  def compile(self, timeout=None, disableAnalytical=False, **kwargs):
    """timeout: give up on a closed-form inverse after so many seconds and use the numeric tables; disableAnalytical: do
    not try (random_number_generator.py:72-119).  timeout None: 2 s like the reference (ODW_ANALYTIC_TIMEOUT overrides).  Other keywords: constants of the density expression."""
    expr, syms, order, used = self._prepare(**kwargs)
    if self._tables is not None and used == self._constantsDict and bool(disableAnalytical) == self._analytic_off:
      return
    self._analytic_off = bool(disableAnalytical)
    if len(order) != 2:
      raise ValueError(f'expected two variables, found {order}')
    e0 = np.linspace(*self._variableDomains[order[0]], self._resolution(order[0], 2))
    e1 = np.linspace(*self._variableDomains[order[1]], self._resolution(order[1], 2))
    m0, m1 = (e0[1:] + e0[:-1]) / 2, (e1[1:] + e1[:-1]) / 2
    self._inverses = None
    if not disableAnalytical:
      from . import analytic
      try:
        self._inverses = analytic.inverses(expr, syms, {n: self._variableDomains[n] for n in order},
                                           timeout=analytic.default_timeout() if timeout is None else timeout)
      except analytic.AnalyticFailure:
        self._inverses = None
    lam = sy.lambdify(syms, expr, modules=['numpy', 'scipy'])
    depends_on_last = syms[1] in expr.free_symbols

    def row(v1):
      p = lam(m0, np.full_like(m0, v1))
      if not hasattr(p, 'shape') or np.shape(p) != m0.shape:
        p = m0 * 0 + p
      if (p < 0).any():
        raise ValueError(f'found negative probability density, expression: {expr}')
      return p

    with np.errstate(invalid='ignore', divide='ignore'):
      return self._compile_tables(expr, syms, order, used, e0, e1, m0, m1, row, depends_on_last)

  def _compile_tables(self, expr, syms, order, used, e0, e1, m0, m1, row, depends_on_last):
    if depends_on_last:
      cdf0 = np.empty((len(m1), len(e0)))
      marg = np.empty(len(m1))
      for i, v in enumerate(m1):
        p = row(v)
        marg[i] = p.sum()
        cdf0[i, 0] = 0.0
        np.cumsum(p, out=cdf0[i, 1:])
        cdf0[i] /= cdf0[i, -1]
    else:
      p = row(m1[0])
      marg = np.full(len(m1), p.sum())
      c = np.concatenate([[0.0], np.cumsum(p)])
      cdf0 = (c / c[-1])[None, :]
    cdf1 = np.concatenate([[0.0], np.cumsum(marg)])
    # integral of the density over its domain (mid-point rule, like the tables): what a discrete event's
    # weight is compared with (freecad_elements/optical_group.py)
    self._mass = float(cdf1[-1]) * float(e0[1] - e0[0]) * float(e1[1] - e1[0])
    cdf1 = cdf1 / cdf1[-1]
    self._mode = 'numeric'
    if self._inverses is not None:
      # analytic mode: the same knots, their cdf values from the closed form (exact where the mid-point sums are 2nd order)
      inv0, inv1 = self._inverses
      try:
        a1 = _as_cdf(inv1.cdf_at(e1))
        if depends_on_last:
          a0 = np.array([_as_cdf(inv0.cdf_at(e0, np.full_like(e0, v))) for v in m1])
        else:
          a0 = _as_cdf(inv0.cdf_at(e0, np.full_like(e0, m1[0])))[None, :]
        cdf0, cdf1 = a0, a1
        self._mode = 'analytic'
      except ValueError:
        self._inverses = None                    # (a closed form that does not evaluate to a cdf on the grid: numeric)
    self._tables = SamplerTables(e0, cdf0, e1, cdf1)
    self._order = order
    self._constantsDict = used

  def tables(self):
    if self._tables is None:
      self.compile()
    return self._tables

  def mass(self):
    """integral of the (unnormalised) density over the domain, for the constants last compiled with"""
    if self._tables is None:
      self.compile()
    return self._mass

  def draw(self, N=None, constants=None):
    """host draw with numpy's global RNG, consuming uniforms in the
    reference's order (random_number_generator.py:492-528): u_last, one
    unused block, u_first, one unused block"""
    if self._tables is None or (constants is not None and constants != self._constantsDict):
      self.compile(**(constants or {}))
    n = 1 if N is None else max(1, int(round(N)))
    u_last = np.random.random_sample(n)
    np.random.random_sample(n)
    u_first = np.random.random_sample(n)
    np.random.random_sample(n)
    if self._inverses is not None:
      v1 = self._inverses[1](u_last)             # (the closed form itself, as the reference's analytic mode draws)
      v0 = self._inverses[0](u_first, v1)
    else:
      v0, v1 = self._tables.draw(u_last, u_first)
    res = {self._order[0]: v0, self._order[1]: v1}
    names = self._variableOrder or self._order
    out = np.array([res[k] for k in names])
    return out if N is not None else out[:, 0]


  def drawPseudo(self, N, bins=None, overdrawFactor=0.1, overdrawIterations=50, constants=None):
    """pseudo-random mode (random_number_generator.py:562-682): N samples whose
    (theta, phi) histogram is thinned towards the expected one.  Each of the
    `overdrawIterations` rounds adds N*overdrawFactor fresh draws, histograms
    everything on a coarse grid and deletes random members of the most
    over-populated bin until only N are left.  Consumes numpy's global RNG
    in the reference's order: the result is the reference's, sample for
    sample, from the same seed (tests/golden/pseudo_draws.npz)."""
    if N <= 1:
      raise ValueError('N must be greater than one in pseudo random mode')
    if overdrawFactor <= 0:
      raise ValueError('overdrawFactor must be greater than zero')
    if overdrawIterations <= 1:
      raise ValueError('overdrawIterations must be greater than one')
    if not self._variableOrder:
      raise ValueError('variableOrder must be passed to constructor to use pseudo random mode.')
    if self._tables is None or (constants is not None and constants != self._constantsDict):
      self.compile(**(constants or {}))
    expr, syms, order, _ = self._prepare(**self._constantsDict)
    by_name = {str(v): v for v in syms}
    expected_fn = sy.lambdify([by_name[n] for n in reversed(self._variableOrder)], expr, modules=['numpy', 'scipy'])
    nvar = len(self._variableOrder)
    pool = None
    for _ in range(round(int(overdrawIterations))):
      fresh = self.draw(N=round(N * overdrawFactor))
      if pool is None:
        # the first round starts from (1 + factor) * N draws; its `fresh` block is discarded
        pool = self.draw(N=round(N * (1 + overdrawFactor)))
      else:
        pool = np.concatenate([pool[..., ~np.isnan(pool[0])], fresh], axis=-1)
      if bins is None:
        bins = int((overdrawFactor * np.sqrt(overdrawIterations) * N)**(1 / (3 * nvar)))
      hist, edges = np.histogramdd(pool.T, bins=bins)
      centers = [(e[1:] + e[:-1]) / 2 for e in edges]
      expected = expected_fn(*np.meshgrid(*reversed(centers)))
      if not hasattr(expected, 'shape'):
        expected = expected * np.ones(hist.shape)
      while True:
        excess = hist / hist.sum() - expected / expected.sum()
        worst = np.argwhere(excess == excess.max())[0]
        inside = None
        for i, (e, b) in enumerate(zip(edges, worst)):
          hit = np.logical_and(e[b] < pool[i], pool[i] <= e[b + 1])
          inside = hit if inside is None else np.logical_and(inside, hit)
        members = np.argwhere(inside)
        if len(members) > 0:
          victim = members[int(np.random.random() * members.shape[0])]
          pool[(Ellipsis,) + tuple(victim)] = np.nan
          hist[tuple(worst)] -= 1
        else:
          pool = pool[..., ~np.isnan(pool[0])]
          pool = pool[..., -int(N):]
          break
        if np.sum(~np.isnan(pool[-1])) <= N:
          break
    result = pool[..., ~np.isnan(pool[0])]
    if pool.shape[-1] / result.shape[-1] > 5:
      import warnings
      warnings.warn('pseudo random generation was not very successful, maybe bins '
                    'or overdraw parameters have to be tweaked...')
    return result[..., -int(round(N)):]


class ScalarRandomVariable:
  '''
  Scalar valued random variable; only `findGrid` (fan mode ray placement,
  random_number_generator.py:685-725 + points_by_density.py:25-38) is needed.
  '''

  def __init__(self, probabilityDensity, variableDomain, variable=None, numericalResolution=None):
    self._probabilityDensity = probabilityDensity
    self._domain = tuple(float(v) for v in variableDomain)
    self._variable = variable
    self._resolution = numericalResolution
    self._expr = None
    self._inverse = None
    self._mode = 'not yet compiled'

  def compile(self, timeout=None, disableAnalytical=False, **constants):
    expr = sy.sympify(self._probabilityDensity)
    for name, val in constants.items():
      if name in [str(s) for s in expr.free_symbols]:
        expr = expr.subs(name, val)
    free = [str(s) for s in expr.free_symbols]
    var = self._variable or (free[0] if free else 'x')
    if len(free) > 1 or (free and var not in free):
      raise ValueError(f'expression "{expr}" seems to have more than one free variable after '
                       f'substituting constants; did you pass all constants to .compile()?')
    l1, l2 = self._domain
    kw = dict(nonnegative=True) if l1 >= 0 else dict(nonpositive=True) if l2 <= 0 else {}
    self._sym = sy.Symbol(var, real=True, **kw)
    self._expr = expr.subs(sy.Symbol(var), self._sym)
    self._inverse = None
    self._mode = 'numeric'
    if not disableAnalytical and np.isfinite(l1) and np.isfinite(l2):
      from . import analytic
      try:
        self._inverse = analytic.inverses(self._expr, [self._sym], {var: (l1, l2)},
                                          timeout=analytic.default_timeout() if timeout is None else timeout)[0]
        self._mode = 'analytic'
      except analytic.AnalyticFailure:
        self._inverse = None

  def tables(self, **constants):
    """numeric-mode inverse-CDF table of the single variable
    (random_number_generator.py:337-369, 372-464 with one variable):
    -> (edges, cdf / cdf[-1])"""
    if self._expr is None or constants:
      self.compile(**constants)
    l1, l2 = self._domain
    if not np.isfinite(l1) or not np.isfinite(l2):
      raise ValueError(f'numerical solution requires finite limits, but found limits [{l1}, {l2}]')
    res = self._resolution if self._resolution else 5 + int(1e6)
    edges = np.linspace(l1, l2, _odd(res))
    if self._inverse is not None:
      try:
        return edges, _as_cdf(self._inverse.cdf_at(edges))     # analytic mode: the closed-form cdf at the same knots
      except ValueError:
        self._inverse, self._mode = None, 'numeric'
    mid = (edges[1:] + edges[:-1]) / 2
    p = sy.lambdify(self._sym, self._expr, modules=['numpy', 'scipy'])(mid)
    if not hasattr(p, 'shape') or np.shape(p) != mid.shape:
      p = mid * 0 + p
    if (p < 0).any():
      raise ValueError(f'found negative probability density, expression: {self._expr}')
    cdf = np.concatenate([[0.0], np.cumsum(p)])
    return edges, cdf / cdf[-1]

  def mode(self):
    return self._mode

  def draw(self, N=None, **constants):
    """host draw with numpy's global RNG in the reference's order
    (random_number_generator.py:492-528): u, one unused block"""
    if self._expr is None or constants:
      self.compile(**constants)
    n = 1 if N is None else max(1, int(round(N)))
    if self._inverse is not None:
      u = np.random.random_sample(n)
      np.random.random_sample(n)
      v = self._inverse(u)
      return v if N is not None else v[0]
    edges, cdf = self.tables()
    u = np.random.random_sample(n)
    np.random.random_sample(n)
    v = np.interp(u, cdf, edges)
    return v if N is not None else v[0]

  def findGrid(self, N, constants=None):
    if self._expr is None or constants:
      # (the grid follows the density itself, not its inverse cdf: no analytic attempt -- the reference makes one here
      #  too, random_number_generator.py:691-692, and waits up to its timeout for every fan without using the outcome)
      self.compile(disableAnalytical=True, **(constants or {}))
    l1, l2 = self._domain
    if not np.isfinite(l1) or not np.isfinite(l2):
      raise ValueError('variable domains must be finite for grid generation')
    res = self._resolution if self._resolution else 5 + int(1e6)
    X = np.linspace(l1, l2, _odd(res))
    Y = sy.lambdify(self._sym, self._expr, modules=['numpy', 'scipy'])(X)
    if not hasattr(Y, 'shape'):
      Y = Y * np.ones(X.shape)
    from .points_by_density import generatePointsWithGivenDensity1D
    pts = generatePointsWithGivenDensity1D((X, Y), N)
    return pts[(pts >= X.min()) & (pts <= X.max())]
