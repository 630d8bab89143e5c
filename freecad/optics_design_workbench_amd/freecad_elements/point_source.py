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
from ..scene.placement import Placement, from_axis_angle


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
  gp = _bake.globalPlacements(doc, obj)[0]
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
