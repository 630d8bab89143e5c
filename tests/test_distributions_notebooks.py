"""The reference's pure-python sampler tests restated (numeric mode, the only
mode of this build): test/10-pure-python-notebooks/
  distributions_quantitative.ipynb  cells 15 (scalar RV histogram rms < 3e-2),
                                    19 (vector RV 2-D histogram rms < 0.1),
                                    21-26 (variable order, symmetric transfer function)
  distributions_basics.ipynb        cells 8, 11, 39, 40 (numeric mode, shapes)
  meshes_by_density.ipynb           cells 8, 13, 16 (findGrid length / symmetry),
                                    29, 33, 37, 41, 45 (generatePointsWithGivenDensity1D)
Analytic-mode and DiracDelta cells (6-13) have no counterpart here."""
import numpy as np
import pytest

from freecad.optics_design_workbench_amd import distributions
from freecad.optics_design_workbench_amd.distributions import (calcDiffDensity, generatePointsWithGivenDensity1D)


@pytest.mark.parametrize('expect, expr, domain', [
  (lambda x: x**2, 'x**2', (0, 10)),
  (lambda x: 0.5 * x + np.cos(x), 'x/2+cos(x)', (0, 4 * np.pi)),
  (lambda x: np.exp(-(x - 3)**2), 'exp(-(x-3)**2)', (1, 5)),
])
@pytest.mark.parametrize('disableAnalytical', [True, False])
def test_scalar_histograms(expect, expr, domain, disableAnalytical):
  # (cell 15 runs every density with and without the analytic mode)
  np.random.seed(1)
  x = distributions.ScalarRandomVariable(expr, variableDomain=domain)
  x.compile(disableAnalytical=disableAnalytical, timeout=20)
  assert x.mode() == ('numeric' if disableAnalytical or expr == 'x/2+cos(x)' else 'analytic')
  H, bins = np.histogram(x.draw(1e6), bins=50)
  bins = (bins[1:] + bins[:-1]) / 2
  want = expect(bins)
  want /= want.max()
  H = H / H.max()
  assert np.sqrt(np.mean(((want - H) / H.max())**2)) < 3e-2


@pytest.mark.parametrize('expect, expr, order, domain', [
  (lambda x, y: np.exp(-(x**2 + y**2)), 'exp(-(x**2 + y**2))', ['x', 'y'], dict(x=(-1, 2), y=(-3, 2))),
  (lambda x, y: np.exp(-x**2), 'exp(-theta**2)', ['theta', 'phi'], dict(theta=(0, np.pi), phi=(0, 2 * np.pi))),
  (lambda x, y: np.exp(-x**2 / (1 + y / 3)**2), 'exp(-theta**2/(1+phi/3)**2)', ['theta', 'phi'],
   dict(theta=(0, np.pi), phi=(0, 2 * np.pi))),
])
@pytest.mark.parametrize('disableAnalytical', [True, False])
def test_vector_histograms(expect, expr, order, domain, disableAnalytical):
  # (cell 19 likewise)
  np.random.seed(2)
  x = distributions.VectorRandomVariable(expr, variableDomains=domain, variableOrder=order)
  x.compile(disableAnalytical=disableAnalytical)
  assert x.mode() in (('numeric',) if disableAnalytical else ('numeric', 'analytic'))
  H, bx, by = np.histogram2d(*x.draw(1e6), bins=(50, 55))
  X, Y = (bx[1:] + bx[:-1]) / 2, (by[1:] + by[:-1]) / 2
  want = expect(*np.meshgrid(X, Y))
  assert np.sqrt(np.mean((want / want.max() - H.T / H.max())**2)) < 0.1


def test_variable_order_and_symmetric_transfer_function():
  x = distributions.VectorRandomVariable('.25-(x-.5)**2 + .02*y', variableOrder=['y', 'x'],
                                         variableDomains=dict(x=(0, 1), y=(0, 3)),
                                         numericalResolutions=dict(x=7, y=9))
  x.compile()
  assert x._order == ['y', 'x']
  d = x.draw(N=5)
  assert d.shape == (2, 5) and np.all((0 <= d[0]) & (d[0] <= 3)) and np.all((0 <= d[1]) & (d[1] <= 1))
  # the marginal of the last variable (x) is symmetric about 0.5: so is its transfer function
  t = x.tables()
  U = np.linspace(0, 1, 100)
  Y = np.interp(U, t.phi_cdf, t.phi_edges)
  assert np.sqrt(np.mean((Y - (Y.max() - Y[::-1]))**2)) < 1e-5
  assert len(t.phi_edges) == 7 and t.t_cdf.shape == (6, 9)


def test_basics_numeric_mode_and_shapes():
  gen = distributions.VectorRandomVariable(
      probabilityDensity='exp(-(theta/(sigma*(1+0.0000000000000000001*phi**2)))**2)',
      variableDomains=dict(theta=(0, np.pi), phi=(0, 2 * np.pi)), variableOrder=['theta', 'phi'])
  gen.compile(sigma=.1)
  assert gen.mode() == 'numeric'
  theta, phi = gen.draw(N=1e5)
  assert len(theta) == int(1e5)
  gen = distributions.VectorRandomVariable(
      probabilityDensity='exp(-(theta/(sigma*(1+0.5*cos(phi + pi/3))))**2)', variableOrder=['theta', 'phi'],
      variableDomains=dict(theta=(0, np.pi), phi=(0, 2 * np.pi)))
  gen.compile(sigma=1)
  assert gen.mode() == 'numeric' and np.shape(gen.draw(N=15)) == (2, 15)
  assert np.shape(distributions.ScalarRandomVariable('cos(x)+2', (0, np.pi)).draw(N=15)) == (15,)
  with pytest.raises(ValueError):        # numeric mode needs finite limits (random_number_generator.py:351-355)
    distributions.VectorRandomVariable('exp(-(theta/sigma)**2)',
                                       variableDomains=dict(theta=(0, np.inf), phi=(0, 2 * np.pi))).compile(sigma=.1)


def test_find_grid_length_and_symmetry():
  srv = distributions.ScalarRandomVariable('exp(-x**2)', variableDomain=(-5, 5))
  X = srv.findGrid(N=51)
  assert len(X) == 51
  assert abs(X[len(X) // 2]) < 1e-9
  assert np.max(np.abs(X + X[::-1])) < 1e-9


@pytest.mark.parametrize('X, Y, N, limit', [
  (np.linspace(-1, 2, 500), lambda X: np.exp(-5 * X**2), 20, 1e-2),
  (np.linspace(-1, 2, 500), lambda X: np.arctan(1e5 * (np.exp(-5 * X**2) - .5)) / np.pi + .5, 10, 1e-5),
  (np.linspace(-1, 3, 500), lambda X: np.arctan(20 * np.exp(-5 * X**2)) * (1 + X), 25, 1e-2),
  (np.linspace(-1, 5, 500), lambda X: np.exp(-5 * X**2) + 0.7 * np.exp(-5 * (X - 2)**2), 50, 1e-2),
  (np.linspace(-1, 5, 500), lambda X: np.exp(-5 * X**2) + 0.7 * np.exp(-5 * (X - 2)**2), 5000, 1e-3),
])
def test_points_with_given_density(X, Y, N, limit):
  Y = Y(X)
  pts = generatePointsWithGivenDensity1D(density=(X, Y), N=N)
  assert pts.shape == (N,)
  dX, dDens = calcDiffDensity(pts)
  errs = [abs((Y / Y.max())[np.argmin(np.abs(X - x))] - y)**2 for x, y in zip(dX, dDens / dDens.max())]
  assert np.sqrt(np.mean(sorted(errs)[2:-2])) < limit
