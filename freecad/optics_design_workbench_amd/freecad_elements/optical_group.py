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
  * both `theta_in` and `theta_refl` on a LENS: the two are tied by Snell's law, theta_refl =
    asin(mu sin theta_in) with mu = n1 / n2 of the hit (ray.py:171-199; total reflection: pi - theta_in),
    and mu takes a handful of values per group (entering from vacuum or from another medium, leaving with
    n2 = 1).  One family over theta_in per value of mu; the device picks the family by the hit's own mu.
Densities with DiracDelta terms (analytic mode, random_number_generator.py:204-320: discrete events
beside a continuum) are split into ATOMS -- c * DiracDelta(theta - a) [* DiracDelta(phi - b)], a linear in
the constants: theta = a with probability c / (sum of the atoms' weights + the continuum's integral), phi = b
or uniform over its domain -- and a continuum that is tabulated as above; the device decides between them
with the uniform it would draw phi from.  `DiracDelta(theta)` [* `DiracDelta(phi)`] as ray *modification*
(the default the reference's GUI writes into new groups) leaves the direction unchanged (theta = 0: both
rotations are identities) and is dropped, like `DiracDelta(theta - theta_refl) * DiracDelta(phi - phi_refl)`
as primary density (the ideal direction with certainty).
"""
from dataclasses import dataclass

import numpy as np
import sympy as sy

from .. import distributions

PRIMARY, MODIFY = 0, 1
AXIS_NONE, AXIS_THETA_IN, AXIS_THETA_REFL = 0, 1, 2
_CONSTANTS = ('theta_in', 'phi_in', 'theta_refl', 'phi_refl')
DEFAULT_FAMILY = 129
SNELL_FAMILY = 65          # members per value of mu of a lens density with both constants
MAX_ATOMS = 4              # ODW_SURF_MAX_ATOMS
MU_ANY, MU_TOTAL_REFLECTION = 0.0, -1.0


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
  mu: float = MU_ANY        # > 0: serves hits with n1 / n2 = mu; -1: total reflection on a lens; 0: every hit
  atom_mass: np.ndarray = None     # (n_family, n_atoms) probability of every discrete event per member
  atom_theta: np.ndarray = None    # (n_atoms, 3) theta = a0 + a1 theta_in + a2 theta_refl
  atom_phi: np.ndarray = None      # (n_atoms, 3) (0, b, b): phi = b; (1, lo, hi): phi uniform over [lo, hi]

  @property
  def n_atoms(self):
    return 0 if self.atom_mass is None else int(np.asarray(self.atom_mass).shape[-1])

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
  """c * DiracDelta(theta - theta_refl) * DiracDelta(phi - phi_refl): the ideal direction with certainty (the
  default the reference's authors have in mind for Reflected/RefractedProbabilityDensity, optical_group.py:39-41,
  50-52).  Without the second factor phi is uniform over its domain -- a cone around the normal: an atom."""
  theta, phi, tr, pr = (sy.Symbol(n) for n in ('theta', 'phi', 'theta_refl', 'phi_refl'))
  d_theta, d_phi = sy.DiracDelta(theta - tr), sy.DiracDelta(phi - pr)
  if not expr.has(d_theta) or not expr.has(d_phi):
    return False
  rest = expr.subs(d_theta, 1).subs(d_phi, 1)
  if rest.has(sy.DiracDelta) or rest.free_symbols:
    return False
  try:
    return bool(rest.evalf() > 0)
  except TypeError:
    return False


def _split_atoms(expr, phi_dom):
  """density = continuum + sum of atoms.  -> (continuum expression or None, [(weight expr, (a0, a1, a2),
  (0, b, b) | (1, phi_lo, phi_hi))]); raises NotImplementedError for anything else with a DiracDelta"""
  theta, phi = sy.Symbol('theta'), sy.Symbol('phi')
  ti, tr = sy.Symbol('theta_in'), sy.Symbol('theta_refl')
  expr = expr.subs({sy.Symbol('phi_in'): 0, sy.Symbol('phi_refl'): 0})
  continuum, atoms = sy.Integer(0), []
  for term in sy.Add.make_args(sy.expand(expr)):
    deltas = [f for f in sy.Mul.make_args(term) if isinstance(f, sy.DiracDelta)]
    if not deltas:
      if term.has(sy.DiracDelta):
        raise NotImplementedError(f'DiracDelta inside "{term}" (powers, function arguments) is not tabulated')
      continuum += term
      continue
    weight = term
    for d in deltas:
      weight = weight / d
    if weight.has(sy.DiracDelta):
      raise NotImplementedError(f'"{term}": a DiracDelta to a power / more than one per variable is not tabulated')
    a = b = None
    for d in deltas:
      arg = d.args[0]
      if len(d.args) > 1 and d.args[1] != 0:
        raise NotImplementedError(f'derivatives of DiracDelta ("{term}") are not tabulated')
      if arg.has(theta) and not arg.has(phi):
        pol = sy.Poly(arg, theta)
        if pol.degree() != 1 or a is not None:
          raise NotImplementedError(f'"{term}": DiracDelta argument not linear in theta, or two of them')
        c1, c0 = pol.all_coeffs()
        a = sy.simplify(-c0 / c1)
        weight = weight / sy.Abs(c1)
      elif arg.has(phi) and not arg.has(theta):
        pol = sy.Poly(arg, phi)
        if pol.degree() != 1 or b is not None:
          raise NotImplementedError(f'"{term}": DiracDelta argument not linear in phi, or two of them')
        c1, c0 = pol.all_coeffs()
        b = sy.simplify(-c0 / c1)
        weight = weight / sy.Abs(c1)
      else:
        raise NotImplementedError(f'"{term}": a DiracDelta has to fix theta or phi')
    if a is None:
      raise NotImplementedError(f'"{term}": a discrete phi with a continuous theta is not tabulated')
    lin = sy.Poly(a, ti, tr)
    if lin.total_degree() > 1:
      raise NotImplementedError(f'"{term}": theta of a discrete event has to be linear in theta_in / theta_refl')
    a_lin = tuple(float(lin.coeff_monomial(m)) for m in (1, ti, tr))
    weight = weight.subs(theta, a)
    if b is not None:
      if b.free_symbols:
        raise NotImplementedError(f'"{term}": phi of a discrete event has to be a number')
      weight = weight.subs(phi, b)
      phi_rule = (0.0, float(b), float(b))
    else:
      if weight.has(phi):
        raise NotImplementedError(f'"{term}": a discrete theta with a phi-dependent weight is not tabulated')
      weight = weight * (phi_dom[1] - phi_dom[0])          # integrated over phi: uniform over its domain
      phi_rule = (1.0, float(phi_dom[0]), float(phi_dom[1]))
    atoms.append((sy.simplify(weight), a_lin, phi_rule))
  if len(atoms) > MAX_ATOMS:
    raise NotImplementedError(f'more than {MAX_ATOMS} discrete events in one density')
  return (continuum if continuum != 0 else None), atoms


def snellConstants(theta_in, mu):
  """theta_refl of the ideal refracted direction for a hit at theta_in with n1 / n2 = mu (snellsLaw, ray.py:488-495);
  mu = MU_TOTAL_REFLECTION: the mirrored direction, pi - theta_in; None where Snell's law has no solution"""
  if mu == MU_TOTAL_REFLECTION:
    return np.pi - theta_in
  x = mu * np.sin(theta_in)
  return float(np.arcsin(x)) if x <= 1.0 else None


def lensMus(scene_iors, g):
  """the values n1 / n2 can take at a hit on lens group g: entering (n2 = its own index) from vacuum or from any
  medium, leaving (n2 = 1 always, ray.py:189) from any medium or from none; + total reflection when mu can exceed 1.
  scene_iors: refractive index of every group that can be a medium (lenses, transmission gratings), by group."""
  n_g = float(scene_iors[g])
  media = sorted({1.0} | {float(v) for v in scene_iors.values()})
  mus = {n1 / n_g for n1 in media} | {n1 for n1 in media}
  out = sorted(mus)
  if any(m > 1.0 for m in out):
    out.append(MU_TOTAL_REFLECTION)
  return out


def _bakeOne(obj, group_index, kind, density, theta_dom, phi_dom, optical_type, n_family, mu=MU_ANY):
  expr = sy.sympify(density)
  names = {str(s) for s in expr.free_symbols}
  atoms = []
  cont_expr = expr
  if expr.has(sy.DiracDelta):
    if kind == MODIFY and _is_identity_dirac(expr):
      return None
    if kind == PRIMARY and _is_ideal_dirac(expr):
      return None        # theta = theta_refl, phi = phi_refl = 0: Rot(n x d, theta_refl) n is the ideal direction
    try:
      cont_expr, atoms = _split_atoms(expr, phi_dom)
    except NotImplementedError as e:
      raise NotImplementedError(f'{obj.Name}: density "{density}": {e}') from None
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
  snell = False
  if 'theta_in' in consts and 'theta_refl' in consts:
    # mirrors: theta_refl = pi - theta_in; lenses: Snell's law for this sampler's mu
    axis, lo, hi = AXIS_THETA_IN, 0.0, np.pi / 2
    snell = optical_type != 'Mirror'
  elif 'theta_in' in consts:
    axis, lo, hi = AXIS_THETA_IN, 0.0, np.pi / 2
  elif 'theta_refl' in consts:
    # specular reflection: theta_refl = pi - theta_in; refraction: [0, pi/2], total reflection beyond
    axis, lo, hi = AXIS_THETA_REFL, (np.pi / 2 if optical_type == 'Mirror' else 0.0), np.pi
  nf = int(n_family) if axis != AXIS_NONE else 1
  vrv = None
  if cont_expr is not None:
    vrv = distributions.VectorRandomVariable(
        probabilityDensity=str(cont_expr) if atoms else density, variableOrder=('theta', 'phi'),
        variableDomains=dict(theta=theta_dom, phi=phi_dom))
  weight_fns = [sy.lambdify([sy.Symbol('theta_in'), sy.Symbol('theta_refl')], w, modules=['numpy']) for w, _, _ in atoms]
  phi_cdf, t_cdf, masses = [], [], []
  phi_edges = t_edges = None
  for k in range(nf):
    c = lo + k * (hi - lo) / (nf - 1) if nf > 1 else lo
    constants = dict(fixed)
    valid = True
    if axis == AXIS_THETA_IN:
      constants['theta_in'] = c
      if 'theta_refl' in consts:
        if snell:
          tr = snellConstants(c, mu)
          valid = tr is not None
          constants['theta_refl'] = tr if valid else 0.0
        else:
          constants['theta_refl'] = np.pi - c    # specular reflection (ray.py:482-486)
    elif axis == AXIS_THETA_REFL:
      constants['theta_refl'] = c
    cont_mass = 0.0
    if vrv is not None and valid:
      # (families of tables, one member per value of the per-hit constant: numeric mode -- an analytic attempt per member
      #  would cost up to its timeout hundreds of times; the reference compiles per hit and may go either way)
      vrv.compile(disableAnalytical=True, **{n: v for n, v in constants.items() if n in {str(x) for x in cont_expr.free_symbols}})
      t = vrv.tables()
      phi_cdf.append(t.phi_cdf.copy())
      t_cdf.append(t.t_cdf.copy())
      phi_edges, t_edges = t.phi_edges, t.t_edges
      cont_mass = vrv.mass() if atoms else 1.0
      if not np.isfinite(cont_mass):
        cont_mass = 0.0
    else:
      phi_cdf.append(None)
      t_cdf.append(None)
    if atoms:
      w = np.array([max(0.0, float(f(constants.get('theta_in', 0.0), constants.get('theta_refl', 0.0)))) for f in weight_fns])
      if not valid:
        w = w * np.nan
      total = w.sum() + (cont_mass if (phi_cdf[-1] is not None and np.isfinite(phi_cdf[-1]).all()) else 0.0)
      masses.append(w / total if total > 0 else w * np.nan)
  if vrv is None:
    # discrete events only: the tables are never read (the atoms' probabilities add up to one)
    phi_edges, t_edges = np.array([phi_dom[0], phi_dom[1]], dtype=np.float64), np.array([theta_dom[0], theta_dom[1]], dtype=np.float64)
    phi_cdf = [np.array([0.0, 1.0]) for _ in range(nf)]
    t_cdf = [np.array([[0.0, 1.0]]) for _ in range(nf)]
  else:
    nanp, nant = np.full(len(phi_edges), np.nan), np.full((1, len(t_edges)), np.nan)
    phi_cdf = [nanp if c is None else c for c in phi_cdf]
    t_cdf = [nant if c is None else c for c in t_cdf]
  if atoms:
    # a member whose continuum vanishes still has its atoms: a flat table stands in for the (unused) continuum
    for k in range(nf):
      if np.isfinite(masses[k]).all() and masses[k].sum() > 1 - 1e-12 and not (np.isfinite(phi_cdf[k]).all() and np.isfinite(t_cdf[k]).all()):
        phi_cdf[k] = np.linspace(0.0, 1.0, len(phi_edges))
        t_cdf[k] = np.linspace(0.0, 1.0, len(t_edges))[None, :]
  # members whose density vanishes on the whole domain (a narrow lobe around a
  # constant far outside the theta domain) have no distribution: 0/0 in the
  # reference too.  They take the tables of the nearest member that has one.
  good = [k for k in range(nf) if np.isfinite(phi_cdf[k]).all() and np.isfinite(t_cdf[k]).all()
          and (not atoms or np.isfinite(masses[k]).all())]
  if not good:
    raise ValueError(f'{obj.Name}: probability density "{density}" vanishes on its whole domain')
  for k in range(nf):
    if k not in good:
      j = min(good, key=lambda g: abs(g - k))
      phi_cdf[k], t_cdf[k] = phi_cdf[j], t_cdf[j]
      if atoms:
        masses[k] = masses[j]
  rows = max(c.shape[0] for c in t_cdf)
  t_cdf = [np.broadcast_to(c, (rows, c.shape[1])) for c in t_cdf]
  return BakedSurfaceSampler(group=group_index, kind=kind, axis=axis, lo=float(lo), hi=float(hi),
                             phi_edges=np.ascontiguousarray(phi_edges), phi_cdf=np.ascontiguousarray(phi_cdf),
                             t_edges=np.ascontiguousarray(t_edges), t_cdf=np.ascontiguousarray(t_cdf),
                             expression=density, mu=float(mu),
                             atom_mass=np.ascontiguousarray(masses, dtype=np.float64) if atoms else None,
                             atom_theta=np.array([a for _, a, _ in atoms], dtype=np.float64) if atoms else None,
                             atom_phi=np.array([r for _, _, r in atoms], dtype=np.float64) if atoms else None)


_CACHE = {}


def _needs_snell_families(density, optical_type):
  if optical_type == 'Mirror':
    return False
  try:
    names = {str(x) for x in sy.sympify(density).free_symbols}
  except (sy.SympifyError, TypeError):
    return False
  return 'theta_in' in names and 'theta_refl' in names


def surfaceSamplers(obj, group_index, n_family=DEFAULT_FAMILY, media=None):
  """tables of one optical group (empty list for ideal surfaces); only mirrors
  and lenses scatter (ray.py:146-199).  media: {group index: refractive index} of every group a ray can be
  inside of (lenses, transmission gratings) -- needed by lens densities that name both theta_in and theta_refl
  (one table family per value n1 / n2 can take, `lensMus`)"""
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
    mus = [MU_ANY]
    family = int(n_family)
    if kind == PRIMARY and _needs_snell_families(dens, t):
      iors = dict(media or {})
      iors.setdefault(group_index, float(obj._props.get('RefractiveIndex', 1.0)))
      mus = lensMus(iors, group_index)
      family = min(family, SNELL_FAMILY)
    for mu in mus:
      key = (kind, dens, td, pd, t, family, float(mu))
      if key not in _CACHE:
        _CACHE[key] = _bakeOne(obj, group_index, kind, dens, td, pd, t, family, mu=mu)
      s = _CACHE[key]
      if s is not None:
        out.append(BakedSurfaceSampler(**{**s.__dict__, 'group': group_index}))
  return out
