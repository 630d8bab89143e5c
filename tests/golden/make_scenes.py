#!/usr/bin/env python3
"""Scene fixtures: the document descriptions of the reference's benchmark /
example / test FCStd files, reduced to what the FCStd-lite loader reads.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_scenes.py

An FCStd file is a zip of data files.  Only `Document.xml` (object list and
property values), the binary `PlacementList*` members (Draft link-array
placements) and the BRep payloads named in BREP (shapes of objects without a
parametric recipe, plus three primitives whose volumes are known in closed
form) are kept; other payloads, GUI state and thumbnails are dropped.
The fixtures are data (scene descriptions), not code.
"""
import os
import zipfile

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scenes')

SCENES = {
  'minimal': 'benchmark/minimal.FCStd',
  'lensesAndMirrors': 'benchmark/lensesAndMirrors.FCStd',
  'lensesAndMirrorsSequential': 'benchmark/lensesAndMirrorsSequential.FCStd',
  'hugeArray': 'benchmark/hugeArray.FCStd',
  'GettingStarted': 'examples/1-getting-started/GettingStarted.FCStd',
  'source-and-absorber': 'test/70-point-source-slow/source-and-absorber.FCStd',
  'gaussian': 'test/50-old-tests/gaussian.FCStd',
  'lens-overlap': 'test/50-old-tests/lens-overlap.FCStd',
  'global-placement-main': 'test/22-global-placement/main.FCStd',
  'nested-structure': 'test/22-global-placement/nested-structure.FCStd',
  'simulation-modes-main': 'test/21-simulation-modes/main.FCStd',
  'mirror-diffuse': 'test/50-old-tests/mirror-diffuse.FCStd',
  'playground': 'test/50-old-tests/playground.FCStd',
  'grating': 'test/50-old-tests/grating.FCStd',
  'edmund-optics-lens': 'test/50-old-tests/edmund-optics-lens.FCStd',
  'mirror': 'test/50-old-tests/mirror.FCStd',
  'imported-stepfile-as-surface-source': 'test/80-surface-source-slow/imported-stepfile-as-surface-source.FCStd',
  'external-file': 'test/22-global-placement/external-file.FCStd',
  'external-file2': 'test/22-global-placement/external-file2.FCStd',
  'lambert-source': 'test/50-old-tests/lambert-source.FCStd',
  'nesting': 'test/50-old-tests/nesting.FCStd',
  'replay': 'test/50-old-tests/replay.FCStd',
}

BREP = {
  'nested-structure': ['Body.Shape.brp', 'Box.Shape.brp', 'Sphere.Shape.brp', 'Cylinder.Shape.brp'],
  'edmund-optics-lens': ['Part__Feature.Shape.brp', 'Part__Feature001.Shape.brp'],
  'mirror': ['Body.Shape.brp'],
  'imported-stepfile-as-surface-source': ['Part__Feature.Shape.brp', 'Part__Feature001.Shape.brp'],
}

if __name__ == '__main__':
  os.makedirs(OUT, exist_ok=True)
  for name, rel in SCENES.items():
    src = os.path.join(REF, rel)
    if not os.path.exists(src):
      print('missing', src)
      continue
    dst = os.path.join(OUT, name + '.FCStd')
    with zipfile.ZipFile(src) as zin, zipfile.ZipFile(dst, 'w', zipfile.ZIP_DEFLATED) as zout:
      for info in zin.infolist():
        if (info.filename == 'Document.xml' or info.filename.startswith('PlacementList')
            or info.filename in BREP.get(name, ())):
          zout.writestr(info.filename, zin.read(info.filename))
    print(name, os.path.getsize(src), '->', os.path.getsize(dst))
