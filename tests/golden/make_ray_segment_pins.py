#!/usr/bin/env python3
"""Ray polylines the reference itself traced (OpenCASCADE intersections) and left in its documents.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_ray_segment_pins.py

When a simulation draws its rays, every ray becomes a `Part::Feature` named
`RaySegment…` whose Shape is a compound of straight edges, one per segment
Ray.traceRay yielded, and the object is made a child of the light source
(generic_source.py:96-138).  Several of the reference's test documents were saved
with such children in place: their `RaySegment*.Shape.brp` payloads hold, with 17
digits, where the reference's own tracing path (Part.Shape intersections, Snell,
mirrors, gratings) took each ray through the very document stored next to them.

The coordinates are those the reference stored: the source's LOCAL frame, since
a line is drawn as `Part.makeLine(gpMi*p1, gpMi*p2)` with gpMi the inverse of the
source's global placement (generic_source.py:107-108).  The test maps them back
with the source's global placement.

Only numbers are kept: per document the name of the owning source and one row
per stored edge  [ray, x1, y1, z1, x2, y2, z2, dx, dy, dz]  (ray = position of
the RaySegment object in the source's ElementList; end points = curve origin +
parameter range * direction, edges in the order the polyline is walked).

Not every document's rays belong to the geometry saved with them:
`test/50-old-tests/mirror.FCStd` holds rays that refract through what is now a
mirror, reflected off a face tilted 42.7 deg where the saved placement says 45 --
drawn before later edits of the document.  It is left out (nothing about it could
be checked); so are stored rays whose shape is empty.
"""
import os
import re
import zipfile

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ray_segments.npz')

DOCS = {
  'grating': 'test/50-old-tests/grating.FCStd',
  'playground': 'test/50-old-tests/playground.FCStd',
  'gaussian': 'test/50-old-tests/gaussian.FCStd',
  'lambert-source': 'test/50-old-tests/lambert-source.FCStd',
  'nesting': 'test/50-old-tests/nesting.FCStd',
  'replay': 'test/50-old-tests/replay.FCStd',
  'edmund-optics-lens': 'test/50-old-tests/edmund-optics-lens.FCStd',
  'lens-overlap': 'test/50-old-tests/lens-overlap.FCStd',
  'mirror-diffuse': 'test/50-old-tests/mirror-diffuse.FCStd',
}


def edges_of(text):
  """[(p1, p2, direction)] of the straight edges of one BRep text, walked from the free start"""
  m = re.search(r'Curves (\d+)\n', text)
  if m is None or int(m.group(1)) == 0:
    return []
  lines = text[m.end():].split('\n')[:int(m.group(1))]
  assert all(l.split()[0] == '1' for l in lines), 'a ray edge that is not a line'
  curves = [np.array(l.split()[1:], dtype=np.float64) for l in lines]
  segs = []
  for e in re.finditer(r'Ed\n [^\n]*\n1\s+(\d+)\s+(\d+)\s+(\S+)\s+(\S+)\n', text):
    c = curves[int(e.group(1)) - 1]
    assert int(e.group(2)) == 0, 'a located edge'
    a, b = float(e.group(3)), float(e.group(4))
    segs.append((c[:3] + a * c[3:], c[:3] + b * c[3:], c[3:]))
  key = lambda p: tuple(np.round(p, 9))
  starts = {key(s[0]): s for s in segs}
  ends = {key(s[1]) for s in segs}
  free = [s for s in segs if key(s[0]) not in ends]
  assert len(free) == 1, 'not one polyline'
  chain = [free[0]]
  while key(chain[-1][1]) in starts and len(chain) < len(segs):
    chain.append(starts[key(chain[-1][1])])
  assert len(chain) == len(segs), 'not one polyline'
  return chain


def owner_lists(xml):
  """{source name: [RaySegment names in ElementList order]}"""
  out = {}
  for m in re.finditer(r'<Object name="([^"]+)"[^>]*>(.*?)</Object>', xml, re.S):
    el = re.search(r'<Property name="ElementList"[^>]*>(.*?)</Property>', m.group(2), re.S)
    if el is None:
      continue
    kids = [v for v in re.findall(r'value="([^"]*)"', el.group(1)) if v.startswith('RaySegment')]
    if kids:
      out[m.group(1)] = kids
  return out


if __name__ == '__main__':
  data = {}
  for name, rel in DOCS.items():
    with zipfile.ZipFile(os.path.join(REF, rel)) as z:
      owners = owner_lists(z.read('Document.xml').decode())
      assert len(owners) == 1, (name, list(owners))
      (source, kids), = owners.items()
      rows = []
      for i, k in enumerate(kids):
        for p1, p2, d in edges_of(z.read(k + '.Shape.brp').decode()):
          rows.append(np.r_[float(i), p1, p2, d])
    rows = np.array(rows)
    data[name + '__source'] = np.array(source)
    data[name + '__edges'] = rows
    print(f'{name:22s} {source:24s} {len(np.unique(rows[:, 0])):4d} rays {len(rows):4d} edges')
  np.savez_compressed(OUT, **data)
  print(OUT, os.path.getsize(OUT), 'bytes')
