#!/usr/bin/env python3
"""Golden vectors for the sampler's ANALYTIC mode from the reference's own module (random_number_generator.py:72-320),
loaded in memory as tests/golden/make_golden.py does.  Run in the authoring container only (needs /root/reference):
    python tests/golden/make_analytic_golden.py
For the densities of test/10-pure-python-notebooks/distributions_quantitative.ipynb cell 15 (scalar) and cell 19 (two
variables), with and without `disableAnalytical`: the mode the reference reports and its transform evaluated on fixed
uniform numbers (the valid root among the solutions, as its draw() selects it) -- numbers only.  -> analytic_modes.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference_modules  # noqa: E402

mods = load_reference_modules()
rng = mods[2]
if not hasattr(rng, 'ScalarRandomVariable'):
  rng = sys.modules['freecad.optics_design_workbench.distributions.random_number_generator']

out = {}
U = np.linspace(0.0, 1.0, 513)[1:-1]          # (the end points: 0 / 0 in some closed forms)
out['u'] = U


def valid_root(transforms, u, params, lo, hi):
  vals = np.array([np.asarray(t(u, *params), dtype=np.complex128) for t in transforms])
  vals = np.where(np.abs(vals.imag) < 1e-12, vals.real, np.nan).astype(np.float64)
  ok = (vals >= lo) & (vals <= hi)
  assert np.all(ok.sum(axis=0) == 1), 'no / more than one valid root'
  return vals[ok.argmax(axis=0), np.arange(vals.shape[1])]


SCALAR = [('x2', 'x**2', (0.0, 10.0)), ('xcos', 'x/2+cos(x)', (0.0, 4 * np.pi)), ('gauss', 'exp(-(x-3)**2)', (1.0, 5.0))]
for tag, expr, dom in SCALAR:
  for disable in (True, False):
    x = rng.ScalarRandomVariable(expr, variableDomain=dom)
    x.compile(disableAnalytical=disable, timeout=20)
    key = f'{tag}_{"numeric" if disable else "auto"}'
    out[key + '_mode'] = np.array(x.mode())
    v = x._vrv if hasattr(x, '_vrv') else x
    transforms, discrete = v._transformLambdas[0]
    assert not discrete
    if v.mode() == 'analytic':
      out[key + '_x'] = valid_root(transforms, U, [], *dom)
    else:
      out[key + '_x'] = np.asarray(transforms[0](U), dtype=np.float64)
    print(key, v.mode(), out[key + '_x'][:3])

VECTOR = [('gxy', 'exp(-(x**2 + y**2))', ['x', 'y'], dict(x=(-1.0, 2.0), y=(-3.0, 2.0))),
          ('gtheta', 'exp(-theta**2)', ['theta', 'phi'], dict(theta=(0.0, np.pi), phi=(0.0, 2 * np.pi)))]
for tag, expr, order, dom in VECTOR:
  for disable in (True, False):
    x = rng.VectorRandomVariable(expr, variableDomains=dom, variableOrder=order)
    x.compile(disableAnalytical=disable, timeout=30)
    key = f'{tag}_{"numeric" if disable else "auto"}'
    out[key + '_mode'] = np.array(x.mode())
    # the last variable first (no parameters), then the first one given the last (as draw() does)
    last_t, _ = x._transformLambdas[1]
    first_t, _ = x._transformLambdas[0]
    if x.mode() == 'analytic':
      v1 = valid_root(last_t, U, [], *dom[order[1]])
      v0 = valid_root(first_t, U[::-1].copy(), [v1], *dom[order[0]])
    else:
      v1 = np.asarray(last_t[0](U), dtype=np.float64)
      v0 = np.asarray(first_t[0](U[::-1].copy(), v1), dtype=np.float64)
    out[key + '_v1'], out[key + '_v0'] = v1, v0
    print(key, x.mode(), v0[:2], v1[:2])

np.savez_compressed(os.path.join(HERE, 'analytic_modes.npz'), **out)
print('wrote analytic_modes.npz:', sorted(out))
