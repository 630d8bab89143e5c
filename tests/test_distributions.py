"""Product-side sampler tables (distributions.VectorRandomVariable) against the
reference's outputs: same tables, same draws, same fan grids."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = sorted(os.path.basename(p)[len('sampler_'):-4] for p in glob.glob(os.path.join(GOLDEN, 'sampler_*.npz')))


def _vrv(g):
  from freecad.optics_design_workbench_amd.distributions import VectorRandomVariable
  return VectorRandomVariable(
      probabilityDensity=str(g['density']), variableOrder=('theta', 'phi'),
      variableDomains=dict(theta=tuple(g['theta_domain']), phi=tuple(g['phi_domain'])),
      numericalResolutions=dict(theta=float(g['theta_res']), phi=float(g['phi_res'])))


@pytest.mark.parametrize('case', CASES)
def test_tables_and_draws(case):
  g = np.load(os.path.join(GOLDEN, f'sampler_{case}.npz'))
  vrv = _vrv(g)
  vrv.compile()
  assert vrv.mode() == 'numeric'
  t = vrv.tables()
  rows, idx = g['row_index'], g['knot_index']
  if t.n_rows == 1:
    # phi-independent density: the reference's rows are all identical
    assert np.all(g['theta_cdf_knots'] == g['theta_cdf_knots'][0])
    assert np.array_equal(t.t_cdf[0][idx], g['theta_cdf_knots'][0])
  else:
    assert t.n_rows == int(g['n_rows'])
    assert np.array_equal(t.t_cdf[np.ix_(rows, idx)], g['theta_cdf_knots'])
  assert np.array_equal(t.t_edges[idx], g['theta_edges_knots'])
  assert np.array_equal(t.phi_cdf, g['phi_cdf'])
  assert np.array_equal(t.phi_edges, g['phi_edges'])
  for seed in (1, 2):
    np.random.seed(seed)
    th, ph = vrv.draw(N=len(g[f'theta_seed{seed}']))
    assert np.array_equal(ph, g[f'phi_seed{seed}'])
    assert np.array_equal(th, g[f'theta_seed{seed}'])


def test_find_grid():
  from freecad.optics_design_workbench_amd.distributions import ScalarRandomVariable
  g = np.load(os.path.join(GOLDEN, 'fan_grid.npz'))
  for name in ('stitched_c1', 'signchange', 'gapped'):
    srv = ScalarRandomVariable(str(g[name + '_density']), tuple(g[name + '_domain']), variable='theta',
                               numericalResolution=float(g[name + '_res']))
    assert np.array_equal(srv.findGrid(N=int(g[name + '_N'])), g[name + '_grid'])


def test_point_source_rv_args():
  """density string handling of PointSourceProxy._rvArgs (point_source.py:277-366)"""
  from conftest import project
  from freecad.optics_design_workbench_amd.freecad_elements import point_source
  proj = project('lensesAndMirrors')
  args = point_source.rvArgs(proj.sourceObject, proj.sourceObject.PowerDensity)
  assert args['variableOrder'] == ('theta', 'phi')
  assert 'Abs(sin(theta))' in args['probabilityDensity'] or 'sin(theta)' in args['probabilityDensity']
  assert args['variableDomains']['theta'] == pytest.approx((0, np.pi / 4))
  assert args['numericalResolutions'] == dict(theta=1e5, phi=1e2)
  g = np.load(os.path.join(GOLDEN, 'sampler_c3_sigma1e-2.npz'))
  t = proj.source.tables
  assert t.n_rows == 1 and len(t.t_edges) == 100001 and len(t.phi_edges) == 101
  assert np.array_equal(t.t_cdf[0][g['knot_index']], g['theta_cdf_knots'][0])
  # infinite focal length -> (r, phi) variables
  p2 = project('source-and-absorber')
  a2 = point_source.rvArgs(p2.sourceObject, p2.sourceObject.PowerDensity)
  assert a2['variableOrder'] == ('r', 'phi')
  with pytest.raises(ValueError):
    point_source.rvArgs(proj.sourceObject, 'exp(-r**2)')   # r forbidden at f=0


@pytest.mark.parametrize('name', ['c3', 'wide'])
def test_draw_pseudo_matches_reference(name):
  """pseudo-random mode (random_number_generator.py:562-682): same samples and
  same RNG consumption as the reference's drawPseudo from the same numpy seed"""
  import json
  from freecad.optics_design_workbench_amd.distributions import VectorRandomVariable
  g = np.load(os.path.join(GOLDEN, 'pseudo_draws.npz'))
  a = json.loads(str(g[name + '_args']))
  vrv = VectorRandomVariable(a['density'], variableOrder=('theta', 'phi'),
                             variableDomains=dict(theta=tuple(a['theta_domain']), phi=tuple(a['phi_domain'])),
                             numericalResolutions=dict(theta=a['theta_res'], phi=a['phi_res']))
  for seed in (3, 4):
    np.random.seed(seed)
    d = vrv.drawPseudo(N=a['N'])
    assert d.shape == (2, a['N'])
    assert np.array_equal(d, g[f'{name}_seed{seed}'])
    assert np.array_equal(np.random.random_sample(4), g[f'{name}_seed{seed}_next'])
  with pytest.raises(ValueError):
    vrv.drawPseudo(N=1)
  with pytest.raises(ValueError):
    vrv.drawPseudo(N=10, overdrawFactor=0)
