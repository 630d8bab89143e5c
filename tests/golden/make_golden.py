#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own Python code.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py

What it does: loads three reference modules *in memory* --
  distributions/random_number_generator.py   (the Monte-Carlo sampler, A5/A6)
  distributions/points_by_density.py          (fan-mode grid helper)
  jupyter_utils/hits.py + histogram.py        (detector binning, A19)
-- under stub parent packages so that the package __init__ (which needs
FreeCAD-side dependencies) is bypassed.  random_number_generator.py uses
three Python-3.11 star-subscripts (lines 438, 650, 651); the text is patched
in memory before exec because this container runs Python 3.10.  No reference
source or bytecode is written anywhere; only inputs and outputs (numpy
arrays) are saved as .npz fixtures next to this script.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference/freecad/optics_design_workbench'
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name):
  m = types.ModuleType(name)
  m.__path__ = []
  sys.modules[name] = m
  return m


def load_reference_modules():
  import matplotlib
  matplotlib.use('Agg')
  for n in ('freecad', 'freecad.optics_design_workbench',
            'freecad.optics_design_workbench.distributions',
            'freecad.optics_design_workbench.jupyter_utils'):
    _stub(n)
  # seaborn is imported by hits.py/histogram.py for plotting only
  if 'seaborn' not in sys.modules:
    try:
      import seaborn  # noqa: F401
    except ImportError:
      sys.modules['seaborn'] = types.ModuleType('seaborn')

  def load(modname, relpath, patch=None):
    full = 'freecad.optics_design_workbench.' + modname
    path = os.path.join(REF, relpath)
    if patch is None:
      spec = importlib.util.spec_from_file_location(full, path)
      mod = importlib.util.module_from_spec(spec)
      sys.modules[full] = mod
      spec.loader.exec_module(mod)
    else:
      src = patch(open(path).read())
      mod = types.ModuleType(full)
      mod.__package__ = full.rsplit('.', 1)[0]
      mod.__file__ = path
      sys.modules[full] = mod
      exec(compile(src, path, 'exec'), mod.__dict__)
    parent, _, leaf = full.rpartition('.')
    setattr(sys.modules[parent], leaf, mod)
    return mod

  io = load('io', 'io.py')
  # logging needs the simulation package (FreeCAD side); discretisation
  # warnings are irrelevant here
  for fn in ('info', 'verb', 'warn', 'err'):
    setattr(io, fn, lambda *a, **k: None)
  pbd = load('distributions.points_by_density', 'distributions/points_by_density.py')

  def patch_rng(s):
    a = '_gridProbsCol = gridProbs[*index,:]'
    b = '_gridProbsCol = gridProbs[tuple(index)+(slice(None),)]'
    assert a in s
    s = s.replace(a, b)
    # lines 650/651 (drawPseudo) use the same 3.11 syntax
    import re
    s, n = re.subn(r'\[\*([A-Za-z_]+),\s*:\]', r'[tuple(\1)+(slice(None),)]', s)
    s, n1 = re.subn(r'\[\.\.\.,\s*\*([A-Za-z_]+)\]', r'[(Ellipsis,)+tuple(\1)]', s)
    s, n2 = re.subn(r'\[\*([A-Za-z_]+)\]', r'[tuple(\1)]', s)
    return s

  rng = load('distributions.random_number_generator',
             'distributions/random_number_generator.py', patch=patch_rng)
  hist = load('jupyter_utils.histogram', 'jupyter_utils/histogram.py')
  hits = load('jupyter_utils.hits', 'jupyter_utils/hits.py')
  return io, pbd, rng, hist, hits


# (name, density in theta/phi AFTER the *abs(sin(theta)) Jacobian of
#  point_source.py:299, theta domain, phi domain, theta res, phi res)
SAMPLER_CASES = [
  ('c3_sigma1e-2', '(exp(-theta**2/(1e-2)**2))*abs(sin(theta))', (0, np.pi / 4), (0, 2 * np.pi), 1e5, 1e2),
  ('c5_sigma0p1', '(exp(-theta^2/0.01))*abs(sin(theta))', (0, np.pi / 4), (0, 2 * np.pi), 1e5, 1e2),
  ('c4_sigma0p2', '(exp(-theta**2/(.2)**2))*abs(sin(theta))', (0, np.pi / 4), (0, 2 * np.pi), 1e5, 1e2),
  ('phidep', '(exp(-theta**2/0.3**2)*(1.5+cos(2*phi)))*abs(sin(theta))', (0, 1.0), (0, 2 * np.pi), 2001, 41),
  ('astig', '(exp(-(theta*cos(phi))**2/0.05**2-(theta*sin(phi))**2/0.2**2))*abs(sin(theta))',
   (0, 0.6), (-np.pi, np.pi), 4001, 61),
]


def make_sampler(rng):
  N = 4096
  for name, dens, tdom, pdom, tres, pres in SAMPLER_CASES:
    vrv = rng.VectorRandomVariable(
        probabilityDensity=dens, variableOrder=('theta', 'phi'),
        variableDomains=dict(theta=tdom, phi=pdom),
        numericalResolutions=dict(theta=tres, phi=pres))
    vrv.compile(disableAnalytical=True)
    assert vrv.mode() == 'numeric'
    out = {}
    for seed in (1, 2):
      np.random.seed(seed)
      th, ph = vrv.draw(N=N)
      np.random.seed(seed)
      u_phi = np.random.random_sample(N)
      _ = np.random.random_sample(N)
      u_th = np.random.random_sample(N)
      out[f'theta_seed{seed}'] = np.asarray(th, dtype=np.float64)
      out[f'phi_seed{seed}'] = np.asarray(ph, dtype=np.float64)
      out[f'u_phi_seed{seed}'] = u_phi
      out[f'u_theta_seed{seed}'] = u_th
    # tables: transform lambdas hold them in closure defaults
    lam_theta = vrv._transformLambdas[0][0][0]
    lam_phi = vrv._transformLambdas[1][0][0]
    kw_t = lam_theta.__kwdefaults__ or {}
    kw_p = lam_phi.__kwdefaults__ or {}
    if not kw_t:
      # defaults are positional-with-default after *params -> keyword-only
      raise RuntimeError('could not read tables from reference closure')
    gp_t = np.array(kw_t['gridProbs'], dtype=np.float64)       # (nphi-1, ntheta) normalised rows that were used
    gp_p = np.array(kw_p['gridProbs'], dtype=np.float64)       # (nphi,)
    edges_t = np.array(kw_t['variableRanges'][0], dtype=np.float64)
    edges_p = np.array(kw_t['variableRanges'][1], dtype=np.float64)
    step = max(1, (gp_t.shape[1] - 1) // 1000)
    idx = np.unique(np.concatenate([np.arange(0, gp_t.shape[1], step), [gp_t.shape[1] - 1]]))
    rows = np.unique(np.concatenate([np.arange(0, gp_t.shape[0], max(1, gp_t.shape[0] // 10)), [gp_t.shape[0] - 1]]))
    # rows of the reference table are normalised lazily, in place, on first
    # use (random_number_generator.py:441); normalise a copy here for all rows
    gp_t_n = gp_t / gp_t[:, -1:]
    out.update(dict(
        density=np.array(dens), theta_domain=np.array(tdom), phi_domain=np.array(pdom),
        theta_res=np.array(tres), phi_res=np.array(pres),
        knot_index=idx, row_index=rows,
        theta_cdf_knots=gp_t_n[np.ix_(rows, idx)], theta_edges_knots=edges_t[idx],
        phi_cdf=gp_p / gp_p[-1], phi_edges=edges_p,
        n_theta_knots=np.array(gp_t.shape[1]), n_rows=np.array(gp_t.shape[0])))
    np.savez_compressed(os.path.join(OUT, f'sampler_{name}.npz'), **out)
    print('sampler', name, 'theta knots', gp_t.shape, 'mean theta', float(np.mean(out['theta_seed1'])))


def make_fan_grid(rng):
  # ScalarRandomVariable.findGrid (fan mode, random_number_generator.py:685-725)
  out = {}
  cases = [
    ('stitched_c1', 'exp(-Abs(theta)**2/(1e-2)**2)', (-np.pi / 4, np.pi / 4), 1e5, 20),
    ('signchange', 'exp(-theta**2/0.1**2)', (-0.3, 0.5), 1e4 + 1, 15),
    ('gapped', 'exp(-theta**2/0.2**2)', (0.05, 0.4), 1e4 + 1, 10),
  ]
  for name, dens, dom, res, N in cases:
    srv = rng.ScalarRandomVariable(probabilityDensity=dens, variable='theta',
                                   variableDomain=dom, numericalResolution=res)
    srv.compile()
    g = np.asarray(srv.findGrid(N=N), dtype=np.float64)
    out[name + '_grid'] = g
    out[name + '_density'] = np.array(dens)
    out[name + '_domain'] = np.array(dom)
    out[name + '_res'] = np.array(res)
    out[name + '_N'] = np.array(N)
    out[name + '_mode'] = np.array(srv.mode())
    print('fan grid', name, srv.mode(), g[:3], g[-3:])
  np.savez_compressed(os.path.join(OUT, 'fan_grid.npz'), **out)


def make_hist(hits_mod):
  rs = np.random.RandomState(7)
  out = {}
  # case A: gaussian spot on a plane z=const, rays along +z
  n = 20000
  pts = np.stack([rs.normal(0.3, 1.0, n), rs.normal(-0.2, 0.5, n), np.full(n, 20.0)], axis=1)
  dirs = np.tile(np.array([0.01, -0.02, 1.0]) / np.linalg.norm([0.01, -0.02, 1.0]), (n, 1))
  # case B: tilted plane, normal (1,1,2)/sqrt6
  nrm = np.array([1.0, 1.0, 2.0]) / np.sqrt(6)
  ex = np.cross(nrm, [0, 0, 1.0]); ex /= np.linalg.norm(ex)
  ey = np.cross(nrm, ex)
  uv = rs.normal(0, 1, (n, 2)) * np.array([2.0, 0.7])
  ptsB = np.array([5.0, -3.0, 40.0]) + uv[:, :1] * ex + uv[:, 1:] * ey
  dirsB = np.tile(nrm, (n, 1)) + rs.normal(0, 0.01, (n, 3))
  for tag, P, D in (('A', pts, dirs), ('B', ptsB, dirsB)):
    for kind, kwargs in (('cart30', dict(bins=30)),
                         ('polar3x50', dict(bins=(3, 50), binCoords='polar')),
                         ('cartlin', dict(bins=[np.linspace(-2, 2, 41), np.linspace(-1, 1, 21)]))):
      h = hits_mod.Hits(dict(points=P.copy(), directions=D.copy(),
                             powers=np.ones(len(P)), isEntering=np.ones(len(P), dtype=int)))
      H = h.histogram(**kwargs)
      key = f'{tag}_{kind}'
      out[key + '_hist'] = np.asarray(H.hist)
      out[key + '_binX'] = np.asarray(H.binX)
      out[key + '_binY'] = np.asarray(H.binY)
      out[key + '_normal'] = np.asarray(H._planeNormal)
      out[key + '_xvec'] = np.asarray(H._xInPlaneVec)
      out[key + '_origin'] = np.asarray(H._origin)
      if kind.startswith('polar'):
        phi, r, dens = H.byAzimuth()
        out[key + '_az_phi'] = phi; out[key + '_az_r'] = r; out[key + '_az_dens'] = dens
        out[key + '_binAreas'] = np.asarray(H.binAreas)
    out[tag + '_points'] = P
    out[tag + '_directions'] = D
  np.savez_compressed(os.path.join(OUT, 'hist_cases.npz'), **out)
  print('hist cases written;', {k: out[k] for k in out if k.endswith('cart30_normal')})


def fwhm_clouds():
  """synthetic spots for the sweep's figure of merit: (points, directions) of a detector plane
  x = const hit by rays along -x (the absorber of examples/1-getting-started), radial profiles
  with a core and a halo so that the log-log fit and the half-maximum search both matter"""
  rs = np.random.RandomState(23)
  out = {}
  for tag, n, core, halo, share in (('tight', 20000, 0.02, 0.5, 0.1), ('wide', 20000, 0.3, 1.5, 0.3),
                                    ('sparse', 3000, 0.05, 0.8, 0.2)):
    k = rs.random_sample(n) < share
    sig = np.where(k, halo, core)
    yz = rs.normal(0, 1, (n, 2)) * sig[:, None] + np.array([0.4, -1.3])
    P = np.concatenate([np.full((n, 1), -41.0), yz], axis=1)
    D = np.tile(np.array([-1.0, 0.0, 0.0]), (n, 1)) + rs.normal(0, 0.02, (n, 3))
    D /= np.linalg.norm(D, axis=1)[:, None]
    out[tag] = (P, D)
  return out


def make_fwhm(hits_mod):
  """the polar histogram `calcFwhm` of examples/1-getting-started/optimize-spotsize.ipynb (cell 8)
  asks the reference's Hits/Histogram for, and the FWHM that cell's arithmetic (numpy only,
  evaluated here on the reference's histogram) gives"""
  out = {}
  for tag, (P, D) in fwhm_clouds().items():
    h = hits_mod.Hits(dict(points=P.copy(), directions=D.copy(), powers=np.ones(len(P)),
                           isEntering=np.ones(len(P), dtype=int)))
    H = h.histogram(binCoords='polar', bins=[np.arange(0, 2 * np.pi, np.pi / 2), np.geomspace(1e-3, 5, 500)])
    phis, r, hists = H.byAzimuth()
    fw = []
    for dens in hists:
      if max(dens) > 0:
        a, b = np.polyfit(np.log(r[dens > 0])[:10], np.log(dens[dens > 0])[:10], deg=1)
        rFit = np.geomspace(min(r), max(r[dens > 10][:10]), 100)
        fit = np.exp(a * np.log(rFit) + b)
        sel = rFit[fit <= max(dens) / 2]
        if len(sel):
          fw.append(min(sel))
    out[tag + '_points'], out[tag + '_directions'] = P, D
    out[tag + '_hist'] = np.asarray(H.hist)
    out[tag + '_dens'] = np.asarray(hists)
    out[tag + '_origin'] = np.asarray(H._origin)
    out[tag + '_normal'] = np.asarray(H._planeNormal)
    out[tag + '_fwhm'] = np.array(np.mean(fw) if fw else np.nan)
  np.savez_compressed(os.path.join(OUT, 'fwhm_cases.npz'), **out)
  print('fwhm cases written;', {k: float(out[k]) for k in out if k.endswith('_fwhm')})


FAN_CASES = {
  # benchmark/minimal.FCStd's source (BASELINE configs[0]: ray-fan mode)
  'c1_minimal': dict(PowerDensity='exp(-theta**2/(1e-2)**2)', FocalLength='0', ThetaDomain='0, pi/4',
                     PhiDomain='0, 2*pi', RadiusDomain='0, 10', Fans=2, RaysPerFan=20, FanPhi0='0',
                     FanModePowerSpan=0.9),
  'gapped': dict(PowerDensity='exp(-theta**2/0.2**2)*(2+cos(phi))', FocalLength='5', ThetaDomain='0.05, 0.3',
                 PhiDomain='0, 2*pi', RadiusDomain='0, 10', Fans=3, RaysPerFan=9, FanPhi0='0.3',
                 FanModePowerSpan=0.8),
  'signchange': dict(PowerDensity='exp(-theta**2/0.1**2)', FocalLength='-20', ThetaDomain='-0.2, 0.3',
                     PhiDomain='-pi, pi', RadiusDomain='0, 10', Fans=2, RaysPerFan=11, FanPhi0='pi/8',
                     FanModePowerSpan=1.0),
  'parallel': dict(PowerDensity='10-abs(r)', FocalLength='inf', ThetaDomain='0, pi/4',
                   PhiDomain='0, 2*pi', RadiusDomain='0, 8', Fans=4, RaysPerFan=7, FanPhi0='0',
                   FanModePowerSpan=0.9),
  'halfphi': dict(PowerDensity='exp(-theta**2/0.05**2)', FocalLength='0', ThetaDomain='0, 0.4',
                  PhiDomain='0, pi/2', RadiusDomain='0, 10', Fans=2, RaysPerFan=8, FanPhi0='0.1',
                  FanModePowerSpan=0.9),
}


def make_fan_rays(io, rng, pbd):
  """PointSourceProxy._generateRays(mode='fans') of the reference with a
  property-bag object instead of a FreeCAD document object; `_makeRay` (which
  needs FreeCAD vectors) is replaced by a recorder, so the golden vectors are
  the (theta|r, phi, fan metadata) sequences the reference would place."""
  base = 'freecad.optics_design_workbench'
  dist = sys.modules[base + '.distributions']
  for k in ('VectorRandomVariable', 'ScalarRandomVariable', 'SampledVectorRandomVariable'):
    setattr(dist, k, getattr(rng, k))
  sim = _stub(base + '.simulation')
  sys.modules[base].simulation = sim
  fe = _stub(base + '.freecad_elements')
  sys.modules[base].freecad_elements = fe

  def load(full, relpath):
    spec = importlib.util.spec_from_file_location(full, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    parent, _, leaf = full.rpartition('.')
    setattr(sys.modules[parent], leaf, mod)
    return mod

  load(base + '.simulation.raytracing_cache', 'simulation/raytracing_cache.py')
  for name in ('find', 'ray'):
    m = types.ModuleType(base + '.freecad_elements.' + name)
    sys.modules[m.__name__] = m
    setattr(fe, name, m)
  load(base + '.freecad_elements.common', 'freecad_elements/common.py')
  load(base + '.freecad_elements.generic_source', 'freecad_elements/generic_source.py')
  ps = load(base + '.freecad_elements.point_source', 'freecad_elements/point_source.py')
  ps.keepGuiResponsiveAndRaiseIfSimulationDone = lambda **kw: None

  class Bag:
    def addProperty(self, *a):
      pass

  out = {}
  for name, props in FAN_CASES.items():
    obj = Bag()
    for k, v in props.items():
      setattr(obj, k, v)
    proxy = ps.PointSourceProxy.__new__(ps.PointSourceProxy)
    rec = []
    proxy._makeRay = lambda obj, thetaOrRadius, phi, power=1, metadata={}, rec=rec: rec.append(
        (float(thetaOrRadius), float(phi), metadata['fanIndex'], metadata['rayIndex'],
         metadata['totalFanCount'], metadata['totalRaysInFan']))
    list(proxy._generateRays(obj, mode='fans'))
    out[name] = np.array(rec, dtype=np.float64)
    out[name + '_props'] = np.array(json.dumps(props))
    print('fan rays', name, len(rec), rec[:2])
  np.savez_compressed(os.path.join(OUT, 'fan_rays.npz'), **out)


# stochastic surfaces: OpticalGroupProxy._getVrv (optical_group.py:212-269)
# builds VectorRandomVariable('('+density+')', variableOrder=('theta','phi'),
# variableDomains=...) with default resolutions and compiles it per hit with
# the constants theta_in, phi_in, theta_refl, phi_refl (:305)
SURFACE_CASES = [
  # test/50-old-tests/mirror-diffuse.FCStd
  ('lambert_mirror', 'cos(theta)**2 * abs(sin(theta))', (-np.pi, -np.pi / 2), (-np.pi, np.pi), [0.3]),
  ('lobe_theta_in', 'exp(-(theta-theta_in)**2/0.05)*(1+0.5*cos(phi-phi_in))', (0, np.pi / 2), (0, 2 * np.pi),
   [0.0, 40 * (np.pi / 2) / 128, np.pi / 2]),
  ('glossy_refl', 'exp(-(theta-theta_refl)**2/0.02)*abs(sin(theta))', (np.pi / 2, np.pi), (-np.pi, np.pi),
   [16 * (np.pi / 2) / 128, 100 * (np.pi / 2) / 128]),
]


def make_surface(rng):
  N = 2048
  out = {}
  for name, dens, tdom, pdom, theta_ins in SURFACE_CASES:
    vrv = rng.VectorRandomVariable(probabilityDensity='(' + dens + ')', variableOrder=('theta', 'phi'),
                                   variableDomains=dict(theta=tdom, phi=pdom))
    out[name + '_density'] = np.array(dens)
    out[name + '_theta_domain'] = np.array(tdom)
    out[name + '_phi_domain'] = np.array(pdom)
    out[name + '_theta_in'] = np.array(theta_ins)
    for j, c in enumerate(theta_ins):
      vrv.compile(disableAnalytical=True, theta_in=c, phi_in=0, theta_refl=np.pi - c, phi_refl=0)
      assert vrv.mode() == 'numeric'
      np.random.seed(11 + j)
      th, ph = vrv.draw(N=N)
      np.random.seed(11 + j)
      u_phi = np.random.random_sample(N)
      _ = np.random.random_sample(N)
      u_th = np.random.random_sample(N)
      out[f'{name}_{j}_theta'] = np.asarray(th, dtype=np.float64)
      out[f'{name}_{j}_phi'] = np.asarray(ph, dtype=np.float64)
      out[f'{name}_{j}_u_phi'] = u_phi
      out[f'{name}_{j}_u_theta'] = u_th
      kw_p = vrv._transformLambdas[1][0][0].__kwdefaults__
      gp_p = np.array(kw_p['gridProbs'], dtype=np.float64)
      out[f'{name}_{j}_phi_cdf'] = gp_p / gp_p[-1]
    print('surface', name, 'mean theta', float(np.mean(out[f'{name}_0_theta'])))
  np.savez_compressed(os.path.join(OUT, 'surface_samplers.npz'), **out)


# densities with DiracDelta terms: the reference's analytic mode with discrete events
# (random_number_generator.py:204-320).  Its draws consume the uniforms in another order than the device's
# sampler, so what is pinned is the distribution: samples of the reference for constants on family knots.
ATOM_CASES = [
  # (name, density, theta domain, phi domain, theta_in) -- mirror: theta_refl = pi - theta_in.  (Continuum parts the
  # reference's symbolic inversion finishes within its two-second deadline: it gives up on abs(), Gaussians ...)
  ('ring_plus_lobe', 'DiracDelta(theta-theta_in) + cos(theta)', (0, np.pi / 2), (0, 2 * np.pi), 40 * (np.pi / 2) / 128),
  ('ring_only', 'DiracDelta(theta-theta_in)', (0, np.pi / 2), (0, 2 * np.pi), 16 * (np.pi / 2) / 128),
  ('two_rings', '0.25*DiracDelta(theta-theta_in) + 0.5*DiracDelta(theta-0.2) + cos(theta)', (0, np.pi / 2), (0, 2 * np.pi),
   100 * (np.pi / 2) / 128),
]


def make_surface_atoms(rng):
  N = 6000
  out = {}
  for name, dens, tdom, pdom, c in ATOM_CASES:
    vrv = rng.VectorRandomVariable(probabilityDensity='(' + dens + ')', variableOrder=('theta', 'phi'),
                                   variableDomains=dict(theta=tdom, phi=pdom))
    vrv.compile(theta_in=c, phi_in=0, theta_refl=np.pi - c, phi_refl=0)
    assert vrv.mode() == 'analytic', vrv.mode()
    np.random.seed(29)
    th, ph = vrv.draw(N=N)
    out[name + '_density'] = np.array(dens)
    out[name + '_theta_domain'] = np.array(tdom)
    out[name + '_phi_domain'] = np.array(pdom)
    out[name + '_theta_in'] = np.array(c)
    out[name + '_theta'] = np.asarray(th, dtype=np.float64)
    out[name + '_phi'] = np.asarray(ph, dtype=np.float64)
    vals, counts = np.unique(np.round(np.asarray(th, dtype=np.float64), 9), return_counts=True)
    print('atoms', name, {float(v): int(k) for v, k in zip(vals, counts) if k > 50})
  np.savez_compressed(os.path.join(OUT, 'surface_atoms.npz'), **out)


# VectorRandomVariable.drawPseudo (random_number_generator.py:562-682) as the
# sources call it (point_source.py:670): N = RaysPerIteration
PSEUDO_CASES = [
  ('c3', '(exp(-theta**2/(1e-2)**2))*abs(sin(theta))', (0, np.pi / 4), (0, 2 * np.pi), 1e5, 1e2, 100),
  ('wide', '(exp(-theta**2/0.3**2)*(1.5+cos(2*phi)))*abs(sin(theta))', (0, 1.0), (0, 2 * np.pi), 2001, 41, 400),
]


def make_pseudo(rng):
  out = {}
  for name, dens, tdom, pdom, tres, pres, N in PSEUDO_CASES:
    vrv = rng.VectorRandomVariable(
        probabilityDensity=dens, variableOrder=('theta', 'phi'),
        variableDomains=dict(theta=tdom, phi=pdom),
        numericalResolutions=dict(theta=tres, phi=pres))
    vrv.compile(disableAnalytical=True)
    for seed in (3, 4):
      np.random.seed(seed)
      draws = np.asarray(vrv.drawPseudo(N=N), dtype=np.float64)
      out[f'{name}_seed{seed}'] = draws
      out[f'{name}_seed{seed}_next'] = np.random.random_sample(4)   # RNG state after the call
    out[name + '_args'] = np.array(json.dumps(dict(density=dens, theta_domain=tdom, phi_domain=pdom,
                                                   theta_res=tres, phi_res=pres, N=N)))
    print('pseudo', name, out[f'{name}_seed3'].shape, out[f'{name}_seed3'][:, :3])
  np.savez_compressed(os.path.join(OUT, 'pseudo_draws.npz'), **out)


# SurfaceSourceProxy draws theta from ScalarRandomVariable(**_rvArgs(..., scalarRandomVar=True))
# (surface_source.py:531, point_source.py:371-386): no Jacobian, one variable
SCALAR_CASES = [
  ('lambert_cos2', 'cos(theta)**2', (0, np.pi / 4), 1e5),          # test/21-simulation-modes/main.FCStd
  ('gauss', 'exp(-theta**2/0.05)', (0, np.pi / 2), 20001),
  ('const', '1', (0.1, 1.2), 1001),
]


def make_scalar(rng):
  out = {}
  N = 4096
  for name, dens, dom, res in SCALAR_CASES:
    srv = rng.ScalarRandomVariable(probabilityDensity=dens, variable='theta', variableDomain=dom,
                                   numericalResolution=res)
    srv.compile(disableAnalytical=True)
    assert srv.mode() == 'numeric'
    np.random.seed(21)
    th = np.asarray(srv.draw(N=N), dtype=np.float64)
    np.random.seed(21)
    u = np.random.random_sample(N)
    out[name + '_theta'] = th
    out[name + '_u'] = u
    out[name + '_args'] = np.array(json.dumps(dict(density=dens, domain=dom, res=res)))
    print('scalar', name, th[:3])
  np.savez_compressed(os.path.join(OUT, 'scalar_draws.npz'), **out)


def fan_math_inputs():
  """synthetic fan-mode hit sets on the plane z = 30: two fans x 21 rays on
  gently bent lines, with duplicates (two hits of one ray), missing rays and,
  in case 'caustic', a fold where the ray order reverses"""
  cases = {}
  rs = np.random.RandomState(21)
  for name in ('regular', 'caustic', 'nozero'):
    pts, ray, fan, tot = [], [], [], []
    for f, ang in enumerate((0.2, 1.9)):
      for i in range(-10, 11):
        if name == 'nozero' and i == 0:
          continue
        if (f, i) in ((0, 7), (1, -3), (1, -2)):
          continue                       # rays that missed the detector
        s = i * 0.8 + 0.01 * i**2
        if name == 'caustic' and i > 5:
          s = 5 * 0.8 + 0.25 - (i - 5) * 0.3   # fold back
        bend = 0.02 * s**2
        x = s * np.cos(ang) - bend * np.sin(ang) + 1.5
        y = s * np.sin(ang) + bend * np.cos(ang) - 0.7
        for rep in range(2 if (f, i) == (0, 3) else 1):
          pts.append([x + 1e-3 * rep, y, 30.0]); ray.append(i); fan.append(f); tot.append(21)
    n = len(pts)
    dirs = np.tile([0.0, 0.0, 1.0], (n, 1)) + rs.normal(0, 1e-3, (n, 3))
    cases[name] = dict(points=np.array(pts), directions=dirs, powers=np.ones(n),
                       isEntering=np.ones(n, dtype=int), rayIndex=np.array(ray), fanIndex=np.array(fan),
                       totalRaysInFan=np.array(tot), totalFanCount=np.full(n, 2))
  return cases


def make_fan_math(hits_mod):
  import warnings
  out = {}
  for name, d in fan_math_inputs().items():
    h = hits_mod.Hits({k: v.copy() for k, v in d.items()})
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      out[name + '_center'] = np.asarray(h.fanCenter())
      out[name + '_centerDists'] = np.asarray(h.fanCenterDists())
      out[name + '_neighborDists'] = np.asarray(h.fanNeighborDists())
      out[name + '_curvs'] = np.asarray(h.fanCurvs())
      out[name + '_missing'] = np.asarray(h.fanMissingRays())
      out[name + '_skipped'] = np.asarray(h.fanSkippedRays())
      out[name + '_healthy'] = np.asarray(h.fanSymmetryHealthy())
      out[name + '_raysPerFan'] = np.asarray(h.raysPerFan())
      out[name + '_fanCount'] = np.asarray(h.fanCount())
      for i, v in h.fanEstimatedPowerDensities().items():
        out[f'{name}_density_{int(i)}'] = np.asarray(v)
      for i, v in h.fanEstimatedCausticIntensities().items():
        out[f'{name}_caustic_{int(i)}'] = np.asarray(v)
      xs = np.linspace(-9, 9, 37)
      for i, fn in h.fanEstimatedPowerDensityFuncs().items():
        out[f'{name}_densityfunc_{int(i)}'] = np.asarray(fn(xs))
      for i, fn in h.fanEstimatedCausticIntensityFuncs().items():
        out[f'{name}_causticfunc_{int(i)}'] = np.asarray([fn(-9.0, 9.0), fn(0.0, 1.0), fn(3.9, 4.1)], dtype=float)
    print('fan math', name, out[name + '_center'], out[name + '_missing'], out[name + '_skipped'],
          out[name + '_healthy'])
  np.savez_compressed(os.path.join(OUT, 'fan_math.npz'), **out)


if __name__ == '__main__':
  io, pbd, rng, hist, hits = load_reference_modules()
  make_sampler(rng)
  make_fan_grid(rng)
  make_hist(hits)
  make_fan_rays(io, rng, pbd)
  make_surface(rng)
  make_surface_atoms(rng)
  make_pseudo(rng)
  make_fan_math(hits)
  make_scalar(rng)
  make_fwhm(hits)
