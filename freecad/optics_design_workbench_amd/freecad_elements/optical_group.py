"""Optical groups: stochastic surface properties -> device sampler tables.

Host-side mirror of `OpticalGroupProxy` (freecad_elements/optical_group.py):
  _getVrv                        :212-269   random variables of the three
                                            probability-density properties
  applyStochasticRayCorrections  :279-323   per-hit compile + draw + rotations
The draw and the rotations run on the device (csrc/odw_kernels.hip `scatter`);
this module prepares their inputs.

The reference compiles the (theta, phi) random variable again at every hit
with the constants `theta_in`, `phi_in = 0`, `theta_refl`, `phi_refl = 0`.
Here the numeric-mode tables are built ahead of the launch:
  * no constant in the expression   -> one table set, exactly the reference's
  * one constant (`theta_in` or `theta_refl`; for mirrors `theta_refl =
    pi - theta_in`, so both may appear) -> a family of `n_family` table sets
    at equidistant values of the constant; the device mixes the two members around the
    hit's constant (member k0 + 1 with probability = the fractional position between the
    knots).  At the family's knots the tables are the reference's, bit for bit.
Densities that only consist of `DiracDelta(theta)` [* `DiracDelta(phi)`] as
ray *modification* (the default the reference's GUI writes into new groups)
leave the direction unchanged (theta = 0: both rotations are identities) and
are dropped.  Other DiracDelta expressions (analytic mode with discrete
events, random_number_generator.py:214-243) are not tabulated -> rejected.
"""
from dataclasses import dataclass

import numpy as np
import sympy as sy

from .. import distributions

PRIMARY, MODIFY = 0, 1
AXIS_NONE, AXIS_THETA_IN, AXIS_THETA_REFL = 0, 1, 2
_CONSTANTS = ('theta_in', 'phi_in', 'theta_refl', 'phi_refl')
DEFAULT_FAMILY = 129


@dataclass
class BakedSurfaceSampler:
  group: int
  kind: int
  axis: int
  lo: float
  hi: float
  phi_edges: np.ndarray     # (n_phi,)
  phi_cdf: np.ndarray       # (n_family, n_phi)
  t_edges: np.ndarray       # (n_t,)
  t_cdf: np.ndarray         # (n_family, rows, n_t)
  expression: str = ''

  @property
  def n_family(self):
    return int(self.phi_cdf.shape[0])

  def member(self, c, u=0.5):
    """index of the family member used for the constant value c (device rule): of the two members
    around c the upper one with probability = the fractional position between the knots, decided by
    the uniform u (the device draws it from Philox counter word 3 = 17 + kind)"""
    if self.axis == AXIS_NONE or self.n_family < 2:
      return 0
    kf = (c - self.lo) * ((self.n_family - 1) / (self.hi - self.lo))
    kf = min(max(kf, 0.0), float(self.n_family - 1))
    k0 = int(np.floor(kf))
    return k0 + (1 if u < kf - k0 else 0)

  def constant(self, k):
    if self.n_family == 1:
      return self.lo
    return self.lo + k * (self.hi - self.lo) / (self.n_family - 1)


def _domain(obj, key, default):
  from .point_source import parsedDomain
  return parsedDomain(obj._props.get(key, default), default)


def _is_identity_dirac(expr):
  """c * DiracDelta(theta) [* DiracDelta(phi)]: theta = 0 with certainty"""
  theta, phi = sy.Symbol('theta'), sy.Symbol('phi')
  rest = expr.subs(sy.DiracDelta(theta), 1).subs(sy.DiracDelta(phi), 1)
  if rest.has(sy.DiracDelta) or theta in rest.free_symbols:
    return False
  if not expr.has(sy.DiracDelta(theta)):
    return False
  try:
    return bool(rest.subs(phi, 0.123).evalf() > 0) if rest.free_symbols <= {phi} else False
  except TypeError:
    return False


def _is_ideal_dirac(expr):
  """c * DiracDelta(theta - theta_refl) [* DiracDelta(phi - phi_refl)]: the ideal
  direction with certainty (the default the reference's authors have in mind for
  Reflected/RefractedProbabilityDensity, optical_group.py:39-41, 50-52)"""
  theta, phi, tr, pr = (sy.Symbol(n) for n in ('theta', 'phi', 'theta_refl', 'phi_refl'))
  d_theta, d_phi = sy.DiracDelta(theta - tr), sy.DiracDelta(phi - pr)
  if not expr.has(d_theta):
    return False
  rest = expr.subs(d_theta, 1).subs(d_phi, 1)
  if rest.has(sy.DiracDelta) or rest.free_symbols:
    return False
  try:
    return bool(rest.evalf() > 0)
  except TypeError:
    return False


def _bakeOne(obj, group_index, kind, density, theta_dom, phi_dom, optical_type, n_family):
  expr = sy.sympify(density)
  names = {str(s) for s in expr.free_symbols}
  if expr.has(sy.DiracDelta):
    if kind == MODIFY and _is_identity_dirac(expr):
      return None
    if kind == PRIMARY and _is_ideal_dirac(expr):
      return None        # theta = theta_refl, phi = phi_refl = 0: Rot(n x d, theta_refl) n is the ideal direction
    raise NotImplementedError(
        f'{obj.Name}: density "{density}" contains DiracDelta terms other than the identity '
        f'modification DiracDelta(theta); discrete events '
        f'(random_number_generator.py:214-243) are not tabulated for the device')
  unknown = names - {'theta', 'phi'} - set(_CONSTANTS)
  if unknown:
    raise ValueError(f'{obj.Name}: variables {sorted(unknown)} exist in expression {density} but are '
                     f'neither theta/phi nor one of {_CONSTANTS}')
  consts = names & set(_CONSTANTS)
  if kind == MODIFY and consts:
    # the reference draws the modification without constants (optical_group.py:317)
    raise ValueError(f'{obj.Name}: variables {sorted(consts)} exist in expression {density} but do not '
                     f'exist in (theta, phi); are all constants specified?')
  fixed = {c: 0.0 for c in ('phi_in', 'phi_refl') if c in consts}
  axis, lo, hi = AXIS_NONE, 0.0, 0.0
  if 'theta_in' in consts and 'theta_refl' in consts:
    if optical_type != 'Mirror':
      raise NotImplementedError(
          f'{obj.Name}: a refracted density depending on both theta_in and theta_refl needs a '
          f'two-parameter table family, which is not built (one constant is supported)')
    axis, lo, hi = AXIS_THETA_IN, 0.0, np.pi / 2
  elif 'theta_in' in consts:
    axis, lo, hi = AXIS_THETA_IN, 0.0, np.pi / 2
  elif 'theta_refl' in consts:
    # specular reflection: theta_refl = pi - theta_in; refraction: [0, pi/2], total reflection beyond
    axis, lo, hi = AXIS_THETA_REFL, (np.pi / 2 if optical_type == 'Mirror' else 0.0), np.pi
  nf = int(n_family) if axis != AXIS_NONE else 1
  vrv = distributions.VectorRandomVariable(
      probabilityDensity=density, variableOrder=('theta', 'phi'),
      variableDomains=dict(theta=theta_dom, phi=phi_dom))
  phi_cdf, t_cdf = [], []
  for k in range(nf):
    c = lo + k * (hi - lo) / (nf - 1) if nf > 1 else lo
    constants = dict(fixed)
    if axis == AXIS_THETA_IN:
      constants['theta_in'] = c
      if 'theta_refl' in consts:
        constants['theta_refl'] = np.pi - c    # specular reflection (ray.py:482-486)
    elif axis == AXIS_THETA_REFL:
      constants['theta_refl'] = c
    vrv.compile(**constants)
    t = vrv.tables()
    phi_cdf.append(t.phi_cdf.copy())
    t_cdf.append(t.t_cdf.copy())
    phi_edges, t_edges = t.phi_edges, t.t_edges
  # members whose density vanishes on the whole domain (a narrow lobe around a
  # constant far outside the theta domain) have no distribution: 0/0 in the
  # reference too.  They take the tables of the nearest member that has one.
  good = [k for k in range(nf) if np.isfinite(phi_cdf[k]).all() and np.isfinite(t_cdf[k]).all()]
  if not good:
    raise ValueError(f'{obj.Name}: probability density "{density}" vanishes on its whole domain')
  for k in range(nf):
    if k not in good:
      j = min(good, key=lambda g: abs(g - k))
      phi_cdf[k], t_cdf[k] = phi_cdf[j], t_cdf[j]
  rows = max(c.shape[0] for c in t_cdf)
  t_cdf = [np.broadcast_to(c, (rows, c.shape[1])) for c in t_cdf]
  return BakedSurfaceSampler(group=group_index, kind=kind, axis=axis, lo=float(lo), hi=float(hi),
                             phi_edges=np.ascontiguousarray(phi_edges), phi_cdf=np.ascontiguousarray(phi_cdf),
                             t_edges=np.ascontiguousarray(t_edges), t_cdf=np.ascontiguousarray(t_cdf),
                             expression=density)


_CACHE = {}


def surfaceSamplers(obj, group_index, n_family=DEFAULT_FAMILY):
  """tables of one optical group (empty list for ideal surfaces); only mirrors
  and lenses scatter (ray.py:146-199)"""
  t = obj._props.get('OpticalType', 'Mirror')
  if t not in ('Mirror', 'Lens'):
    return []
  prim_key = 'ReflectedProbabilityDensity' if t == 'Mirror' else 'RefractedProbabilityDensity'
  specs = []
  dens = str(obj._props.get(prim_key, '') or '').strip()
  if dens:
    specs.append((PRIMARY, '(' + dens + ')', _domain(obj, 'PowerThetaDomain', '-pi/2, pi/2'),
                  _domain(obj, 'PowerPhiDomain', '0, 2*pi')))
  dens = str(obj._props.get('RayModificationProbabilityDensity', '') or '').strip()
  if dens:
    specs.append((MODIFY, dens, _domain(obj, 'ModifyThetaDomain', '-pi/2, pi/2'),
                  _domain(obj, 'ModifyPhiDomain', '0, 2*pi')))
  out = []
  for kind, dens, td, pd in specs:
    key = (kind, dens, td, pd, t, int(n_family))
    if key not in _CACHE:
      _CACHE[key] = _bakeOne(obj, group_index, kind, dens, td, pd, t, n_family)
    s = _CACHE[key]
    if s is not None:
      out.append(BakedSurfaceSampler(**{**s.__dict__, 'group': group_index}))
  return out
