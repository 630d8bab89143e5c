"""Pins of the TRACING path: what the reference's own FreeCAD / OpenCASCADE run left behind.

Build container only (reads /root/reference, never travels to the GPU box):

    python tests/golden/make_notebook_pins.py      ->  tests/golden/notebook_pins.npz

The reference ships no known-answer vectors for intersections, but its example notebooks
carry STORED CELL OUTPUTS that the real `ray.py` on top of OpenCASCADE produced on
examples/1-getting-started/GettingStarted.FCStd:

* visualize-power-density.ipynb
    cell 6  (PNG)  cartesian histogram, bins linspace(-.05, .05, 100), LogNorm colour scale, with the
                   title `Histogram.plot` prints (histogram.py:130-135): plane normal, projected x, origin
    cell 8  (PNG)  polar profile: density over r per azimuth bin, bins [arange(0, 2pi, pi/2), linspace(0, .05, 500)],
                   log-log axes (document saved with Sphere.Radius = 10: optimize-spotsize.ipynb cell 1
                   "restore default value")
* optimize-spotsize.ipynb
    cell 10 (PNG)  spot FWHM over lens radius: 30 radii linspace(9, 11, 30), EndAfterRays = 1e3 each (cell 9),
                   `calcFwhm` of cell 8
    cell 11        wrote radii[argmin(fwhms)] back: the shipped GettingStarted.FCStd holds that radius

This script digitises the curves from the stored images (axes are located by their tick marks, whose
labels are written in the images and restated below; curves by the line colours of matplotlib's default
cycle) and stores NUMBERS only: no pixel data, no notebook text.  The title of cell 6 is rendered text;
it is restated here as read from the image:

    plane normal = [1.00, -0.00, 0.00],  projected x = [0.00, 1.00, 0.00],  origin = [-2.14e-05, 1.00e+01]
"""
import base64
import io
import json
import os
import re
import zipfile

import numpy as np
from PIL import Image

REF = '/root/reference/examples/1-getting-started'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'notebook_pins.npz')

C0 = (31, 119, 180)      # matplotlib default cycle: first line  (#1f77b4)
C3 = (214, 39, 40)       # fourth line (#d62728): cell 8 plots density, fit, fwhm per azimuth -> 2nd density is C3


def cell_png(notebook, cell):
  nb = json.load(open(os.path.join(REF, notebook)))
  for o in nb['cells'][cell].get('outputs', []):
    if 'data' in o and 'image/png' in o['data']:
      return np.asarray(Image.open(io.BytesIO(base64.b64decode(o['data']['image/png']))).convert('RGB')).astype(int)
  raise KeyError((notebook, cell))


def frame(im, min_len=200):
  """the axes' black frame: (left, right, top, bottom) pixel indices of the long dark lines"""
  dark = im.sum(axis=2) < 120
  rows = [i for i, n in enumerate(dark.sum(axis=1)) if n > min_len]
  cols = [i for i, n in enumerate(dark.sum(axis=0)) if n > min_len]
  return cols[0], cols[1], rows[0], rows[1]


def ticks_below(im, bottom, left, right, depth):
  """x pixel of every tick mark that reaches `depth` pixels below the bottom frame line"""
  dark = im.sum(axis=2) < 200
  return [c for c in range(left, right + 1) if dark[bottom + 1:bottom + 1 + depth, c].all()]


def ticks_left(im, left, top, bottom, depth):
  dark = im.sum(axis=2) < 200
  return [r for r in range(top, bottom + 1) if dark[r, left - depth:left].all()]


def colour_rows(im, colour, col, tol=40):
  """rows of the pixels of column `col` that have the line's colour"""
  d = np.abs(im[:, col, :] - np.array(colour)).sum(axis=1)
  return np.nonzero(d < tol)[0]


def fwhm_curve():
  """optimize-spotsize.ipynb cell 10: linear axes; x tick labels 9.00 ... 11.00 (step 0.25), y tick labels
  0.00 ... 0.04 (step 0.01); one line through the 30 (radius, fwhm) points"""
  im = cell_png('optimize-spotsize.ipynb', 10)
  left, right, top, bottom = frame(im)
  xt = ticks_below(im, bottom, left, right, 4)
  yt = ticks_left(im, left, top, bottom, 4)
  assert len(xt) == 9 and len(yt) == 5, (xt, yt)
  px = np.polyfit(xt, np.arange(9.0, 11.01, 0.25), 1)          # pixel -> mm
  py = np.polyfit(yt, np.arange(0.04, -0.001, -0.01), 1)
  assert np.abs(np.polyval(px, xt) - np.arange(9.0, 11.01, 0.25)).max() < 3e-3
  radii = np.linspace(9, 11, 30)
  cols = np.rint((radii - px[1]) / px[0]).astype(int)
  def centre_line(c_from, c_to):
    """stroke centre (row) per column over a stretch that lies inside ONE segment of the polyline"""
    cs, rs = [], []
    for c in range(c_from, c_to + 1):
      rows = colour_rows(im, C0, c)
      if len(rows):
        cs.append(c); rs.append(0.5 * (rows.min() + rows.max()))
    return np.array(cs), np.array(rs)

  fw = []
  for k, c in enumerate(cols):
    # the polyline has a vertex at this column: where the centre lines of the two segments that meet here
    # (fitted 2 ... 7 px away from the vertex; the points are 15.5 px apart) cross the column
    est = []
    if k > 0:
      cs, rs = centre_line(c - 7, c - 2)
      est.append(np.polyval(np.polyfit(cs, rs, 1), c))
    if k < len(cols) - 1:
      cs, rs = centre_line(c + 2, c + 7)
      est.append(np.polyval(np.polyfit(cs, rs, 1), c))
    fw.append(np.polyval(py, np.mean(est)))
  return radii, np.array(fw), abs(py[0])        # (mm per pixel: the digitisation's resolution)


def polar_profile():
  """visualize-power-density.ipynb cell 8: log-log axes; major x ticks 1e-4, 1e-3, 1e-2, major y ticks
  1e5 ... 1e10; densities of the azimuth bins phi = 0.25 pi (C0) and 0.75 pi (C3); 499 radial bins of
  linspace(0, .05, 500)"""
  im = cell_png('visualize-power-density.ipynb', 8)
  left, right, top, bottom = frame(im)
  xt = ticks_below(im, bottom, left, right, 4)          # majors only reach 4 px
  yt = [r for r in ticks_left(im, left, top, bottom, 4)]
  assert len(xt) == 3 and len(yt) == 6, (xt, yt)
  px = np.polyfit(xt, [-4.0, -3.0, -2.0], 1)           # pixel -> log10 r
  py = np.polyfit(yt, [10.0, 9.0, 8.0, 7.0, 6.0, 5.0], 1)
  r_edges = np.linspace(0, .05, 500)
  r = 0.5 * (r_edges[1:] + r_edges[:-1])
  prof = {}
  for name, colour in (('phi025', C0), ('phi075', C3)):
    cols, top_row, mid_row = [], [], []
    for c in range(left + 1, right):
      rows = colour_rows(im, colour, c, tol=30)
      rows = rows[(rows > top) & (rows < bottom)]
      if len(rows):
        cols.append(c); top_row.append(rows.min()); mid_row.append(np.median(rows))
    cols = np.array(cols)
    prof[name] = (10.0**np.polyval(px, cols), 10.0**np.polyval(py, np.array(top_row)),
                  10.0**np.polyval(py, np.array(mid_row)))
  return r, prof, abs(px[0])


def disc_image():
  """visualize-power-density.ipynb cell 6: cartesian histogram, bins linspace(-.05, .05, 100), LogNorm; x tick
  labels -0.06 ... 0.06 (step 0.02), y tick labels -0.04 ... 0.04 (step 0.02).  Bins without hits are white
  (log of 0 is masked), bins with hits are coloured: the spot's bright rim and centre = the green-to-yellow pixels -> the
  disc's extent along x and y"""
  im = cell_png('visualize-power-density.ipynb', 6)
  dark = im.sum(axis=2) < 120
  rows = [i for i, n in enumerate(dark.sum(axis=1)) if n > 300]
  cols = [i for i, n in enumerate(dark.sum(axis=0)) if n > 300]
  left, right, top, bottom = cols[0], cols[1], rows[0], rows[-1]
  xt = ticks_below(im, bottom, left, right, 4)
  yt = ticks_left(im, left, top, bottom, 4)
  assert len(xt) == 7 and len(yt) == 5, (xt, yt)
  px = np.polyfit(xt, np.arange(-0.06, 0.061, 0.02), 1)
  py = np.polyfit(yt, np.arange(0.04, -0.041, -0.02), 1)
  # the LogNorm colour scale (viridis): the rim of the filled disc (the caustic) holds > 10 % of the peak density and is drawn
  # green (green channel > 120, like the centre), bins beyond its edge hold single hits (< 1 %: purple) or none (white)
  inside = im[top + 1:bottom, left + 1:right]
  bright = (inside[..., 1] > 120) & (inside[..., 1] - inside[..., 2] > 20)      # (not grey, not white)
  cs = np.nonzero(bright.sum(axis=0) >= 3)[0] + left + 1
  rs = np.nonzero(bright.sum(axis=1) >= 3)[0] + top + 1
  xr = (np.polyval(px, cs.max()), np.polyval(px, cs.min()))
  yr = (np.polyval(py, rs.min()), np.polyval(py, rs.max()))
  # brightest pixel (yellow = the maximum of the LogNorm scale): the spot centre
  yellow = np.abs(im - np.array([253, 231, 37])).sum(axis=2) < 60
  yy, xx = np.nonzero(yellow[top + 1:bottom, left + 1:right])
  centre = (np.polyval(px, xx.mean() + left + 1), np.polyval(py, yy.mean() + top + 1))
  return np.array(xr), np.array(yr), np.array(centre), abs(px[0])


def shipped_radius():
  """Sphere.Radius stored in the shipped GettingStarted.FCStd (optimize-spotsize.ipynb cell 11 wrote
  radii[argmin(fwhms)] into the file)"""
  xml = zipfile.ZipFile(os.path.join(REF, 'GettingStarted.FCStd')).read('Document.xml').decode()
  obj = xml[xml.index('<Object name="Sphere"', xml.index('<ObjectData')):]
  m = re.search(r'<Property name="Radius"[^>]*>\s*<Float value="([^"]+)"', obj)
  return float(m.group(1))


def main():
  radii, fwhm, fw_res = fwhm_curve()
  r, prof, lg_res = polar_profile()
  xr, yr, centre, img_res = disc_image()
  # --- features of the polar profile --------------------------------------------------
  feats = {}
  for name, (rr, top, mid) in prof.items():
    sel = rr > 0.02
    k = np.argmax(top[sel])                      # the caustic spike: highest point of the curve beyond r = 0.02
    r_peak = rr[sel][k]
    plateau = np.median(mid[(rr > 0.015) & (rr < 0.028)])
    beyond = mid[(rr > r_peak * 1.03) & (rr < 0.048)]
    feats[name] = (r_peak, top[sel][k], plateau, np.median(beyond), beyond.max())
    print(name, 'caustic peak at r = %.5f mm, peak %.3g, plateau %.3g, beyond: median %.3g max %.3g' % feats[name])
  stored = shipped_radius()
  print('fwhm curve:', np.round(fwhm, 4))
  print('argmin radius', radii[np.argmin(fwhm)], 'shipped Sphere.Radius', stored, '= linspace(9,11,30)[12]:',
        np.isclose(stored, radii[12], atol=1e-12))
  print('disc: x', xr, 'y', yr, 'centre', centre, 'resolution', img_res)
  np.savez(
    OUT,
    # optimize-spotsize.ipynb cell 10 / 11
    sweep_radii=radii, sweep_fwhm=fwhm, sweep_fwhm_resolution=fw_res, shipped_sphere_radius=stored,
    sweep_rays_per_run=1e3,
    # visualize-power-density.ipynb cell 6 title (restated from the rendered text)
    title_plane_normal=np.array([1.00, -0.00, 0.00]), title_projected_x=np.array([0.00, 1.00, 0.00]),
    title_origin=np.array([-2.14e-05, 1.00e+01]),
    # cell 6 image: the filled disc's extent along the projected axes, the brightest bin
    disc_x_extent=xr, disc_y_extent=yr, disc_brightest=centre, disc_resolution=img_res,
    # cell 8: digitised curves (r, density at the top of the stroke, density at its middle) and their features
    profile_phi025=np.stack(prof['phi025']), profile_phi075=np.stack(prof['phi075']),
    profile_log10_per_pixel=lg_res,
    caustic_phi025=np.array(feats['phi025']), caustic_phi075=np.array(feats['phi075']),
    profile_bins=np.array([0.0, 0.05, 500]),
  )
  print('wrote', OUT)


if __name__ == '__main__':
  main()
