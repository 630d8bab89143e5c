"""Pins of the oracle's sampler against the reference's own outputs
(tests/golden/sampler_*.npz, produced by tests/golden/make_golden.py from the
reference's random_number_generator.py) and against numpy itself."""
import glob
import os
import types

import numpy as np
import pytest

from conftest import GOLDEN

CASES = sorted(os.path.basename(p)[len('sampler_'):-4] for p in glob.glob(os.path.join(GOLDEN, 'sampler_*.npz')))


def test_philox_known_answers(oracle):
  # Random123 kat_vectors, philox4x32 10 rounds
  assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
  assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
      [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_interp_matches_numpy(oracle):
  rs = np.random.RandomState(3)
  xp = np.concatenate([[0.0], np.cumsum(rs.rand(500) ** 8)])
  xp[100:110] = xp[100]            # flat stretch (duplicate knots)
  xp = np.sort(xp) / xp.max()
  fp = np.linspace(-1, 2, len(xp))
  x = np.concatenate([rs.rand(5000), xp[::7], [0.0, 1.0, -0.1, 1.1]])
  assert np.array_equal(oracle.interp(x, xp, fp), np.interp(x, xp, fp))


def _tables(case):
  from oracle import sampler_oracle
  g = np.load(os.path.join(GOLDEN, f'sampler_{case}.npz'))
  t = sampler_oracle.build_tables(str(g['density']), ('theta', 'phi'),
                                  dict(theta=tuple(g['theta_domain']), phi=tuple(g['phi_domain'])),
                                  dict(theta=float(g['theta_res']), phi=float(g['phi_res'])))
  return g, t


@pytest.mark.parametrize('case', CASES)
def test_tables_match_reference(case):
  g, t = _tables(case)
  cdf0 = t['cdf'][0]
  assert cdf0.shape == (int(g['n_rows']), int(g['n_theta_knots']))
  rows, idx = g['row_index'], g['knot_index']
  norm = cdf0 / cdf0[:, -1:]
  assert np.array_equal(norm[np.ix_(rows, idx)], g['theta_cdf_knots'])
  assert np.array_equal(t['edges'][0][idx], g['theta_edges_knots'])
  assert np.array_equal(t['cdf'][1] / t['cdf'][1][-1], g['phi_cdf'])
  assert np.array_equal(t['edges'][1], g['phi_edges'])


@pytest.mark.parametrize('case', CASES)
def test_draw_matches_reference(case, oracle):
  """numpy restatement AND the C oracle reproduce the reference's draws bit
  for bit from the same uniforms"""
  from oracle import sampler_oracle
  g, t = _tables(case)
  cdf0 = t['cdf'][0] / t['cdf'][0][:, -1:]
  same_rows = bool(np.all(cdf0 == cdf0[0]))
  src = types.SimpleNamespace(
      xform=np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0]), focal_length=0.0, wavelength=500.0, power=1.0,
      tables=types.SimpleNamespace(t_edges=t['edges'][0], t_cdf=cdf0[:1] if same_rows else cdf0,
                                   phi_edges=t['edges'][1], phi_cdf=t['cdf'][1] / t['cdf'][1][-1]))
  for seed in (1, 2):
    u_phi, u_t = g[f'u_phi_seed{seed}'], g[f'u_theta_seed{seed}']
    th, ph = sampler_oracle.draw_from_uniforms(t, u_phi, u_t)
    assert np.array_equal(ph, g[f'phi_seed{seed}'])
    assert np.array_equal(th, g[f'theta_seed{seed}'])
    th_c, ph_c = oracle.sample_uniforms(src, u_phi, u_t)
    assert np.array_equal(ph_c, g[f'phi_seed{seed}'])
    assert np.array_equal(th_c, g[f'theta_seed{seed}'])


def test_find_grid_matches_reference():
  from oracle import sampler_oracle
  g = np.load(os.path.join(GOLDEN, 'fan_grid.npz'))
  for name in ('stitched_c1', 'signchange', 'gapped'):
    got = sampler_oracle.find_grid(str(g[name + '_density']), 'theta', tuple(g[name + '_domain']),
                                   float(g[name + '_res']), int(g[name + '_N']))
    assert np.array_equal(got, g[name + '_grid'])
