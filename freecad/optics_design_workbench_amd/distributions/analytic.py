"""Closed-form inverse CDFs: the sampler's *analytic mode*.

The reference first asks sympy for the inverse of the cumulative distribution of every variable and falls back to its
numeric tables when that fails or takes longer than `timeout` seconds (distributions/random_number_generator.py:72-119,
`_generateAnalyticScalarLambda` :204-320).  The same recipe, restated:

  variable i of (v_0 ... v_{n-1}), drawn after the variables behind it (the last one first):
    p_i(v_i | v_{i+1} ...)  =  the density integrated over the full domains of v_0 ... v_{i-1}
    F_i(x)                  =  integral_{l1}^{x} p_i  /  integral_{l1}^{l2} p_i
    x                       =  the root of F_i(x) = u inside [l1, l2]     (u uniform in [0, 1))

`solve` may return several branches; exactly one lies in the domain for a given u (the reference's draw() checks the
same, :503-509).  Densities with DiracDelta / Heaviside terms (discrete events) are left to the callers that handle them
(freecad_elements/optical_group.py: the surface samplers' atoms); here they make the analytic attempt fail, i.e.
numeric mode -- unless the caller handles them itself.

What the device gets in analytic mode is still a table -- its sampler interpolates (cdf, edge) knots --, but one built
from the closed form: the knots stand at the numeric mode's edges, their cdf values are F_i(edge) exactly instead of
mid-point sums (`cdf_at`), so the only error left is the interpolation between knots 1e-5 of the domain apart."""
import signal
import threading
import time

import numpy as np
import sympy as sy


class AnalyticFailure(Exception):
  pass


def default_timeout():
  """seconds an attempt may take: the reference's 2 (random_number_generator.py:72), or ODW_ANALYTIC_TIMEOUT (test suites
  that compile dozens of densities sympy never inverts set it lower: the outcome for those is the same, sooner)"""
  import os
  try:
    return float(os.environ.get('ODW_ANALYTIC_TIMEOUT', '2'))
  except ValueError:
    return 2.0


class _Deadline:
  """raises in the main thread when the time is up (sympy's integrate / solve cannot be interrupted otherwise); in other
  threads there is no such means: the attempt is not made at all (numeric mode)"""

  def __init__(self, seconds):
    self.seconds = float(seconds)
    self.armed = False

  def __enter__(self):
    if threading.current_thread() is not threading.main_thread():
      raise AnalyticFailure('analytic mode is attempted in the main thread only (it needs an alarm to give up)')
    if self.seconds <= 0:
      raise AnalyticFailure('time is up')
    self.previous = signal.signal(signal.SIGALRM, self._fire)
    signal.setitimer(signal.ITIMER_REAL, self.seconds)
    self.armed = True
    return self

  def _fire(self, *a):
    raise AnalyticFailure('time is up')

  def __exit__(self, *exc):
    if self.armed:
      signal.setitimer(signal.ITIMER_REAL, 0)
      signal.signal(signal.SIGALRM, self.previous)
    return False


def _lambdify(args, expr):
  return sy.lambdify(args, expr, modules=['numpy', 'scipy'])


class Inverse:
  """the closed-form transform of one variable: branches x = g_k(u, later variables...), the cdf F(x, later...)"""

  def __init__(self, var, domain, later, branches, cdf_expr, density_expr):
    self.var, self.domain, self.later = var, domain, list(later)
    self.expressions = list(branches)
    self.cdf_expr, self.density_expr = cdf_expr, density_expr
    u = sy.Symbol('__y', real=True, nonnegative=True)
    self._u = u
    self._branches = [_lambdify([u] + self.later, b) for b in branches]
    x = sy.Symbol('__x', real=True)
    self._cdf = _lambdify([x] + self.later, cdf_expr)

  def __call__(self, u, *later):
    """the root inside the domain for every u (arrays broadcast); ValueError where none or several are"""
    u = np.asarray(u, dtype=np.float64)
    lo, hi = self.domain
    vals = []
    with np.errstate(all='ignore'):
      for f in self._branches:
        v = np.asarray(f(u, *later), dtype=np.complex128)
        v = np.broadcast_to(v, u.shape)
        vals.append(np.where(np.abs(v.imag) <= 1e-12 * np.maximum(1.0, np.abs(v.real)), v.real, np.nan))
    vals = np.array(vals, dtype=np.float64)
    ok = (vals >= lo) & (vals <= hi)
    if not np.all(ok.sum(axis=0) == 1):
      raise ValueError(f'no/more than one valid value found in domain for variable {self.var}')
    return np.take_along_axis(vals, ok.argmax(axis=0)[None, ...], axis=0)[0]

  def cdf_at(self, x, *later):
    with np.errstate(all='ignore'):
      v = np.asarray(self._cdf(np.asarray(x, dtype=np.float64), *later), dtype=np.complex128)
    return np.broadcast_to(v.real, np.shape(x)).astype(np.float64)


_CACHE = {}


def inverses(expr, variables, domains, timeout=2.0):
  """[Inverse per variable] of the density `expr` over `variables` (sympy symbols, in drawing order: the LAST is drawn
  first) with `domains` {name: (l1, l2)}, or AnalyticFailure.  Outcomes are remembered per (expression, domains): the
  second source with the same density does not wait for sympy again -- nor for its timeout."""
  key = (sy.srepr(expr), tuple(str(v) for v in variables), tuple((str(v), tuple(map(float, domains[str(v)]))) for v in variables))
  hit = _CACHE.get(key)
  if hit is not None:
    if isinstance(hit, Exception):
      raise AnalyticFailure(str(hit))
    return hit
  t_end = time.time() + float(timeout)
  try:
    if expr.has(sy.DiracDelta) or expr.has(sy.Heaviside):
      raise AnalyticFailure('discrete events (DiracDelta / Heaviside) are the caller\'s')
    out = []
    for i, var in enumerate(variables):
      with _Deadline(t_end - time.time()):
        p = expr
        for j in range(i):                       # the variables in front of this one: integrated out
          l1, l2 = domains[str(variables[j])]
          p = sy.integrate(p, (variables[j], l1, l2))
        l1, l2 = domains[str(var)]
        if not (np.isfinite(l1) and np.isfinite(l2)):
          raise AnalyticFailure('infinite domains are not supported')
        kw = dict(positive=True) if l1 >= 0 else dict(negative=True) if l2 <= 0 else {}
        x = sy.Symbol('__x', real=True, **kw)
        u = sy.Symbol('__y', real=True, nonnegative=True)
        total = sy.integrate(p, (var, l1, l2))
        partial = sy.integrate(p, (var, l1, x))
        if total.has(sy.Integral) or partial.has(sy.Integral) or total == 0:
          raise AnalyticFailure(f'no closed form for the integral of {p} over {var}')
        cdf = partial / total
        if not cdf.has(x):
          raise AnalyticFailure('the distribution has no continuous part')
        roots = sy.solve(sy.Eq(cdf, u), x, simplify=False)
        if not roots:
          raise AnalyticFailure(f'{cdf} = u is not solvable for {var}')
        later = list(variables[i + 1:])
        inv = Inverse(var, (float(l1), float(l2)), later, roots, cdf.subs(x, sy.Symbol('__x', real=True)), p / total)
        # the reference's probe: ten draws must come out as numbers (random_number_generator.py:101-109)
        probe_later = [np.full(10, 0.5 * (domains[str(v)][0] + domains[str(v)][1])) for v in later]
        probe = inv(np.linspace(0.05, 0.95, 10), *probe_later)
        if not np.all(np.isfinite(probe)):
          raise AnalyticFailure('analytic mode was not successful')
        out.append(inv)
  except AnalyticFailure as e:
    _CACHE[key] = e
    raise
  except Exception as e:                         # whatever sympy raises on the way: numeric mode
    err = AnalyticFailure(f'{type(e).__name__}: {e}')
    _CACHE[key] = err
    raise err
  _CACHE[key] = out
  return out
