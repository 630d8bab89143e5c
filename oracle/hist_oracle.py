"""numpy restatement of the reference's post-hoc detector binning (A19).

TEST INFRASTRUCTURE ONLY (see oracle/odw_oracle.c header for the rule).

Follows
  jupyter_utils/hits.py:62-94     planeProject3dPoints
  jupyter_utils/hits.py:96-174    detectPlaneNormal
  jupyter_utils/hits.py:176-193   histogram
  jupyter_utils/histogram.py:24-89,150-161   Histogram.__init__, byAzimuth
Pinned by tests/golden/hist_cases.npz (outputs of the reference's own classes
on synthetic point clouds, tests/golden/make_golden.py).
"""
import numpy as np

_AXES = [np.array([1, 0, 0]), np.array([0, 1, 0]), np.array([0, 0, 1])]


def detect_plane_normal(points, directions, is_entering, max_points=300, angle_tol=1e-9):
  if np.sum(is_entering == 0) < .51 * len(is_entering):
    directions = directions[is_entering != 0]
  cp = points[::1 + int(points.shape[0] / max_points)]
  cd = directions[::1 + int(directions.shape[0] / max_points)]
  phis = np.linspace(0, np.pi, 30)
  dphi = phis[1] - phis[0]
  thetas = np.linspace(-np.pi / 2, np.pi / 2, 30)
  dtheta = thetas[1] - thetas[0]
  while True:
    cands, spans = [], []
    for phi, theta in zip(*[g.flatten() for g in np.meshgrid(phis, thetas)]):
      n = np.array([np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)])
      p = np.dot(cp, n)
      cands.append((phi, theta))
      spans.append(p.max() - p.min())
    phi_o, theta_o = cands[np.argmin(spans)]
    phis = np.linspace(phi_o - 1.1 * dphi, phi_o + 1.1 * dphi, 10)
    dphi = phis[1] - phis[0]
    thetas = np.linspace(theta_o - 1.1 * dtheta, theta_o + 1.1 * dtheta, 10)
    dtheta = thetas[1] - thetas[0]
    if dphi < angle_tol and dtheta < angle_tol:
      break
  normal = np.array([np.cos(phi_o) * np.sin(theta_o), np.sin(phi_o) * np.sin(theta_o), np.cos(theta_o)])
  proj = np.dot(cd, normal)
  if np.quantile(proj, 0.1) > 0:
    normal = -normal
  elif np.quantile(proj, 0.9) < 0:
    pass
  elif np.quantile(proj, 0.5) < 0:
    normal = -normal
  proj_y = sorted([np.cross(normal, a) for a in _AXES], key=lambda v: -np.linalg.norm(v))[0]
  xvec = np.cross(normal, proj_y)
  if np.sum(xvec) < 0:
    xvec = -xvec
  return normal, xvec


def plane_project(points, normal, xvec):
  X = np.dot(points, xvec / np.linalg.norm(xvec))
  py = np.cross(normal, xvec)
  Y = np.dot(points, py / np.linalg.norm(py))
  return X, Y


def histogram(points, directions, is_entering, bin_coords='cartesian', origin=None, **kwargs):
  normal, xvec = detect_plane_normal(points, directions, is_entering)
  X, Y = plane_project(points, normal, xvec)
  if origin is None:
    origin = np.array([np.median(X), np.median(Y)])
  X = X - origin[0]
  Y = Y - origin[1]
  out = dict(normal=normal, xvec=xvec, origin=origin)
  if bin_coords == 'cartesian':
    out['hist'], out['binX'], out['binY'] = np.histogram2d(X, Y, **kwargs)
  else:
    H, bx, by = np.histogram2d(np.arctan2(X, Y), np.sqrt(X**2 + Y**2), **kwargs)
    phi1, phi2, r1, r2 = bx[:-1], bx[1:], by[:-1], by[1:]
    (r1, phi1), (r2, phi2) = np.meshgrid(r1, phi1), np.meshgrid(r2, phi2)
    areas = (phi2 - phi1) * (r1 + r2) / 2 * (r2 - r1)
    out.update(hist=H, binX=bx, binY=by, binAreas=areas,
               az_phi=(bx[1:] + bx[:-1]) / 2, az_r=(by[:-1] + by[1:]) / 2,
               az_dens=np.array([s for s in (H / areas).T.T]))
  return out
