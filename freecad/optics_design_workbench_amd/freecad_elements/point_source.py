"""Point light source: density string -> sampler tables -> device source.

Host-side mirror of `PointSourceProxy` (freecad_elements/point_source.py):
  _rvArgs        :277-366   Jacobian + r/x/y substitution
  _getVrv        :371-386   cached random variable
  _makeRay       :411-460   (theta|r, phi) -> origin/direction (host copy,
                            used for fan mode; the device has its own)
  _generateRays  :474-656   fan mode ray placement
The Monte-Carlo branch (:659-679) runs on the device; this module only
prepares its inputs.
"""
from dataclasses import dataclass

import numpy as np
import sympy as sy

from .. import distributions
from ..scene import bake as _bake


def parsedDomain(domain, default='0,1'):
  """GenericFreecadElementProxy._parsedDomain (common.py:293-361), parsing
  part: 'a, b' with sympy expressions -> (float, float)"""
  try:
    vals = [float(sy.sympify(d).evalf()) for d in str(domain).split(',')]
    if len(vals) != 2:
      raise ValueError(domain)
  except Exception:
    vals = [float(sy.sympify(d).evalf()) for d in default.split(',')]
  l1, l2 = vals
  return (l2, l1) if l1 > l2 else (l1, l2)


_FORBIDDEN_AT_F0 = ('exp', 'arcsin', 'arccos', 'arctan', 'arctan2', 'arccot', 'arsinh', 'arcosh',
                    'artanh', 'arcoth', 'DiracDelta', 'Piecewise', 'Heaviside', 'True', 'False')


def rvArgs(obj, densityString, variableDomain=None, scalarRandomVar=False):
  """kwargs of the random variable for `densityString` (point_source.py:277-366)"""
  f = float(obj._props.get('FocalLength', 1))
  theta_dom = parsedDomain(obj._props.get('ThetaDomain', '0, pi/4'))
  phi_dom = parsedDomain(obj._props.get('PhiDomain', '0, 2*pi'))
  r_dom = parsedDomain(obj._props.get('RadiusDomain', '0, 10'))
  if np.isfinite(f):
    if np.isclose(f, 0):
      stripped = densityString
      for w in _FORBIDDEN_AT_F0:
        stripped = stripped.replace(w, '')
      for c in 'rxy':
        if c in stripped:
          raise ValueError(f'Variable {c} in power density expression {densityString} '
                           f'is forbidden if focal length is zero .')
    if not scalarRandomVar:
      densityString = '(' + densityString + ')*abs(sin(theta))'
    fs = f'{abs(f):.8e}'
    expr = (sy.sympify(densityString)
            .subs('r', sy.sympify(f'(tan(theta)*{fs})'))
            .subs('x', sy.sympify(f'(tan(theta)*cos(phi)*{fs})'))
            .subs('y', sy.sympify(f'(tan(theta)*sin(phi)*{fs})')))
    if scalarRandomVar:
      return dict(probabilityDensity=str(expr), variable='theta', variableDomain=variableDomain,
                  numericalResolution=float(obj._props.get('ThetaResolutionNumericMode', '1e5')))
    return dict(probabilityDensity=str(expr), variableOrder=('theta', 'phi'),
                variableDomains=dict(theta=theta_dom, phi=phi_dom),
                numericalResolutions=dict(
                    theta=float(obj._props.get('ThetaResolutionNumericMode', '1e5')),
                    phi=float(obj._props.get('PhiResolutionNumericMode', '1e2'))))
  if not scalarRandomVar:
    densityString = '(' + densityString + ')*abs(r)'
  if 'theta' in densityString:
    raise ValueError(f'Variable theta in power density expression {densityString} '
                     f'is forbidden if focal length is infinite.')
  expr = (sy.sympify(densityString)
          .subs('x', sy.sympify('(r*cos(phi))'))
          .subs('y', sy.sympify('(r*sin(phi))')))
  if scalarRandomVar:
    return dict(probabilityDensity=str(expr), variable='r', variableDomain=variableDomain,
                numericalResolution=float(obj._props.get('RadiusResolutionNumericMode', '1e5')))
  return dict(probabilityDensity=str(expr), variableOrder=('r', 'phi'),
              variableDomains=dict(r=r_dom, phi=phi_dom),
              numericalResolutions=dict(
                  r=float(obj._props.get('RadiusResolutionNumericMode', '1e5')),
                  phi=float(obj._props.get('PhiResolutionNumericMode', '1e2'))))


@dataclass
class BakedSource:
  xform: np.ndarray          # (12,) local -> global rows (R|t)
  focal_length: float
  wavelength: float
  power: float
  tables: 'distributions.SamplerTables'
  name: str = ''
  label: str = ''
  rays_per_iteration_scale: float = 1.0


_VRV_CACHE = {}


def getVrv(obj):
  """cached VectorRandomVariable of a source (point_source.py:371-386); the
  cache key covers every property the tables depend on"""
  key = (id(obj._doc), obj.Name) + tuple(
      str(obj._props.get(k)) for k in ('PowerDensity', 'FocalLength', 'ThetaDomain', 'PhiDomain',
                                       'RadiusDomain', 'ThetaResolutionNumericMode',
                                       'RadiusResolutionNumericMode', 'PhiResolutionNumericMode'))
  vrv = _VRV_CACHE.get(key)
  if vrv is None:
    vrv = distributions.VectorRandomVariable(**rvArgs(obj, obj.PowerDensity))
    vrv.compile()
    _VRV_CACHE[key] = vrv
    obj._props['RandomNumberGeneratorMode'] = vrv.mode()
  return vrv


def bakeSource(doc, obj):
  if obj.ProxyClass != 'PointSourceProxy':
    raise NotImplementedError(f'{obj.Name}: {obj.ProxyClass} is outside the accelerated path '
                              f'(SURVEY 8f N4)')
  # _getCoordinateTransformMatricesWithoutLinks (common.py:279-280)
  gp = _bake.globalPlacements(doc, obj, ignoreLinks=True)[0]
  return BakedSource(xform=gp.rows12(), focal_length=float(obj.FocalLength),
                     wavelength=float(obj._props.get('Wavelength', 500)), power=1.0,
                     tables=getVrv(obj).tables(), name=obj.Name,
                     label=obj._props.get('Label', obj.Name),
                     rays_per_iteration_scale=float(obj._props.get('RaysPerIterationScale', 1)))


def makeRay(baked, thetaOrRadius, phi):
  """host copy of _makeRay (point_source.py:411-460) for explicit initial
  conditions (fan mode): -> origin(3), direction(3) in global coordinates"""
  f = baked.focal_length
  if np.isfinite(f):
    theta = thetaOrRadius
    st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    ldir = np.array([st * sp, -st * cp, ct])
    lorg = (np.array([0, 0, 1.0]) - ldir) * f
  else:
    ldir = np.array([0, 0, 1.0])
    lorg = np.array([thetaOrRadius * np.cos(phi), -thetaOrRadius * np.sin(phi), 0.0])
  m = np.asarray(baked.xform).reshape(3, 4)
  p1 = m[:, :3] @ lorg + m[:, 3]
  p2 = m[:, :3] @ (lorg + ldir / np.linalg.norm(ldir)) + m[:, 3]
  d = p2 - p1
  return p1, d / np.linalg.norm(d)


def _fanMode(l1, l2):
  if (l1 > 0 and l2 > 0) or (l1 < 0 and l2 < 0):
    return 'gapped'
  if l1 == 0 or l2 == 0:
    return 'stitched'
  if l1 < 0 and l2 > 0:
    return 'theta-sign-change'
  raise ValueError(f'{l1=}, {l2=}')


def generateFanAngles(obj, maxFanCount=np.inf, maxRaysPerFan=np.inf):
  """fan-mode ray placement of PointSourceProxy._generateRays
  (point_source.py:474-656): -> list of (thetaOrRadius, phi, metadata) in the
  reference's emission order.  One fan covers both the phi and phi+pi side of
  the optical axis; depending on the theta (radius) domain the two sides are
  placed independently ('gapped'), from one symmetric density ('stitched') or
  through the sign of theta ('theta-sign-change')."""
  p = obj._props
  finite = np.isfinite(float(p.get('FocalLength', 1)))
  var = 'theta' if finite else 'r'
  l1, l2 = parsedDomain(p.get('ThetaDomain', '0, pi/4')) if finite else parsedDomain(p.get('RadiusDomain', '0, 10'))
  phiL1, phiL2 = parsedDomain(p.get('PhiDomain', '0, 2*pi'))
  raysPerFan = min(int(p.get('RaysPerFan', 20)), maxRaysPerFan)
  totalFanCount = int(min(int(p.get('Fans', 2)), maxFanCount))
  fanMode = _fanMode(l1, l2)
  if fanMode == 'gapped':
    raysPerFan = max(4, int(np.ceil(raysPerFan / 2) * 2))
  density = p.get('PowerDensity', 'exp(-theta^2/0.01)')
  span = float(p.get('FanModePowerSpan', 0.9))
  phi0 = float(sy.sympify(p.get('FanPhi0', '0')).evalf())
  out = []

  def in_domain(cands):
    return [c for c in cands if phiL1 - 1e-9 <= c <= phiL2 + 1e-9]

  def srv(dens, domain):
    return distributions.ScalarRandomVariable(**rvArgs(obj, dens, variableDomain=domain, scalarRandomVar=True))

  for fanIndex, _phi in enumerate(phi0 + np.linspace(0, np.pi, totalFanCount + 1)[:-1]):
    cands = in_domain(np.arange(_phi - 30 * np.pi, _phi + 31 * np.pi, np.pi))
    if not cands:
      continue
    phiA = cands[int(np.argmin(np.abs(_phi - np.array(cands))))]
    cands = in_domain(np.arange(phiA + np.pi - 30 * np.pi, phiA + np.pi + 31 * np.pi, 2 * np.pi))
    phiB = cands[int(np.argmin(np.abs(phiA + np.pi - np.array(cands))))] if cands else np.nan
    piecewise = f'Piecewise( ( ({phiA}), ({var})>0 ), ( ({phiB}),  True     ) )'

    # restrict the fan to the central FanModePowerSpan of the emitted power;
    # like in the reference the narrowed limits carry over to the next fans
    if 0 < span < 1:
      power = sy.lambdify(var, sy.sympify(density).subs('theta', 'abs(theta)').subs('phi', piecewise))
      limit = max(abs(l1), abs(l2))
      grid = np.linspace(-limit, limit, int(1e5))
      cum = np.cumsum(power(grid) * np.ones_like(grid))
      cum = cum / cum.max()
      a = grid[int(np.argmin(np.abs(cum - (1 - span) / 2)))]
      b = grid[int(np.argmin(np.abs(cum - (1 - (1 - span) / 2))))]
      maxL = max(abs(a), abs(b))
      if abs(l1) > maxL:
        l1 = np.sign(l1) * maxL
      if abs(l2) > maxL:
        l2 = np.sign(l2) * maxL

    side2 = []
    if fanMode == 'gapped':
      rv = srv(density, (l1, l2))
      side1 = rv.findGrid(N=raysPerFan // 2, constants=dict(phi=phiA))
      side2 = rv.findGrid(N=raysPerFan // 2, constants=dict(phi=phiB))
    elif fanMode == 'stitched':
      limit = max(abs(l1), abs(l2))
      expr = sy.sympify(density).subs('theta', 'abs(theta)').subs('r', 'abs(r)')
      if np.isfinite(phiB):
        rv = srv(str(expr.subs('phi', piecewise)), (-limit, limit))
      else:
        rv = srv(str(expr), (0, limit))
      side1 = rv.findGrid(N=raysPerFan, constants=dict(phi=phiA))
    else:
      side1 = srv(density, (l1, l2)).findGrid(N=raysPerFan, constants=dict(phi=phiA))

    if len(side2) > 0:
      side1, side2 = sorted(side1, key=abs), sorted(side2, key=abs)
      idx1 = list(1 + np.arange(len(side1)))
      idx2 = list(-(1 + np.arange(len(side2))))
    else:
      side1 = np.array(sorted(side1))
      idx1 = list(np.arange(len(side1)) - int(np.argmin(np.abs(side1))))
      idx2 = []
    packed = list(zip(idx1, side1, [phiA] * len(side1))) + list(zip(idx2, side2, [phiB] * len(side2)))
    for rayIndex, value, phi in sorted(packed, key=lambda e: abs(e[0]) - .1):
      out.append((float(value), float(phi),
                  dict(fanIndex=int(fanIndex), rayIndex=int(rayIndex), totalFanCount=int(totalFanCount),
                       totalRaysInFan=len(packed))))
  return out


def generateFanRays(obj, baked, **kwargs):
  """-> list of (origin, direction, metadata): explicit initial conditions for
  Tracer.traceRays (the reference's useInitialConditions path)"""
  rays = []
  for value, phi, meta in generateFanAngles(obj, **kwargs):
    o, d = makeRay(baked, value, phi)
    meta = dict(meta, initPhi=phi, initTheta=value if np.isfinite(baked.focal_length) else np.nan)
    rays.append((o, d, meta))
  return rays
