"""Detector binning (A19): oracle restatement and the product's Hits/Histogram
against outputs of the reference's own classes (tests/golden/hist_cases.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

KINDS = {'cart30': dict(bins=30), 'polar3x50': dict(bins=(3, 50), binCoords='polar'),
         'cartlin': dict(bins=[np.linspace(-2, 2, 41), np.linspace(-1, 1, 21)])}


@pytest.fixture(scope='module')
def golden():
  return np.load(os.path.join(GOLDEN, 'hist_cases.npz'))


@pytest.mark.parametrize('tag', ['A', 'B'])
@pytest.mark.parametrize('kind', list(KINDS))
def test_oracle_hist(golden, tag, kind):
  from oracle import hist_oracle
  P, D = golden[f'{tag}_points'], golden[f'{tag}_directions']
  kw = dict(KINDS[kind])
  bc = kw.pop('binCoords', 'cartesian')
  out = hist_oracle.histogram(P, D, np.ones(len(P), dtype=int), bin_coords=bc, **kw)
  key = f'{tag}_{kind}'
  assert np.array_equal(out['normal'], golden[key + '_normal'])
  assert np.array_equal(out['xvec'], golden[key + '_xvec'])
  assert np.array_equal(out['origin'], golden[key + '_origin'])
  assert np.array_equal(out['hist'], golden[key + '_hist'])
  assert np.array_equal(out['binX'], golden[key + '_binX'])
  assert np.array_equal(out['binY'], golden[key + '_binY'])
  if bc == 'polar':
    assert np.array_equal(out['binAreas'], golden[key + '_binAreas'])
    assert np.array_equal(out['hist'] / out['binAreas'], golden[key + '_az_dens'])


@pytest.mark.parametrize('tag', ['A', 'B'])
@pytest.mark.parametrize('kind', list(KINDS))
def test_product_hits_histogram(golden, tag, kind):
  from freecad.optics_design_workbench_amd.jupyter_utils import Hits
  P, D = golden[f'{tag}_points'], golden[f'{tag}_directions']
  h = Hits(dict(points=P.copy(), directions=D.copy(), powers=np.ones(len(P)),
                isEntering=np.ones(len(P), dtype=int)))
  H = h.histogram(**KINDS[kind])
  key = f'{tag}_{kind}'
  # same plane, same origin, same counts as the reference's own classes, bit for bit
  assert np.array_equal(H._planeNormal, golden[key + '_normal'])
  assert np.array_equal(H._xInPlaneVec, golden[key + '_xvec'])
  assert np.array_equal(H._origin, golden[key + '_origin'])
  assert np.array_equal(H.hist, golden[key + '_hist'])
  assert np.array_equal(H.binX, golden[key + '_binX'])
  assert np.array_equal(H.binY, golden[key + '_binY'])
  if kind.startswith('polar'):
    phi, r, dens = H.byAzimuth()
    assert np.array_equal(dens, golden[key + '_az_dens'])
    assert np.array_equal(H.binAreas, golden[key + '_binAreas'])
