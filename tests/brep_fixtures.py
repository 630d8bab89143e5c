"""BRep payloads written by hand for the tests ("CASCADE Topology V1", the text FreeCAD stores for shapes without a
parametric recipe).  The reference's own files hold no surface of revolution of a parabola (README: "slotted
parabolic mirrors" belong to an example that is not in the snapshot), so the blank of a parabolic mirror --
Part::Revolution of a parabola about its axis, closed by a plane -- is spelled out here the way
BRepTools_ShapeSet writes shapes: curves / surfaces by kind number (GeomTools_CurveSet: 4 = parabola "P N Dx Dy
focal"; GeomTools_SurfaceSet: 7 = revolution "P D <curve>"), TShapes numbered from the last to the first."""
import numpy as np


def paraboloid_solid(f=2.5, h=4.0):
  """solid x^2 + y^2 <= 4 f z, z <= h: paraboloid face (seam + rim + degenerated vertex edge) and a planar cap"""
  f, h = float(f), float(h)
  r = float(2.0 * np.sqrt(f * h))
  tp = float(2.0 * np.pi)
  return f"""CASCADE Topology V1, (c) Matra-Datavision
Locations 0
Curve2ds 5
1 0 0 0 1 
1 {tp!r} 0 0 1 
1 0 {r!r} 1 0 
2 0 0 1 0 -0 1 {r!r}
1 0 0 1 0 
Curves 2
4 0 0 0 0 1 0 0 0 1 1 0 0 {f!r}
2 0 0 {h!r} 0 0 1 1 0 -0 -0 1 0 {r!r}
Polygon3D 0
PolygonOnTriangulations 0
Surfaces 2
7 0 0 0 0 0 1 
4 0 0 0 0 1 0 0 0 1 1 0 0 {f!r}
1 0 0 {h!r} 0 0 1 1 0 -0 -0 1 0 
Triangulations 0

TShapes 11
Ve
1e-07
0 0 0
0 0

0101101
*
Ve
1e-07
{r!r} 0 {h!r}
0 0

0101101
*
Ed
 1e-07 1 1 0
1  1 0 0 {r!r}
3  1 2CN 1 0 0 {r!r}
0

0101000
+11 0 -10 0 *
Ed
 1e-07 1 1 0
1  2 0 0 {tp!r}
2  3 1 0 0 {tp!r}
2  4 2 0 0 {tp!r}
0

0101000
+10 0 -10 0 *
Ed
 1e-07 1 1 1
2  5 1 0 0 {tp!r}
0

0101000
+11 0 -11 0 *
Wi

0101100
+9 0 +8 0 -9 0 -7 0 *
Fa
0  1e-07 1 0

0101000
+6 0 *
Wi

0101100
-8 0 *
Fa
0  1e-07 2 0

0101000
+4 0 *
Sh

0101100
+5 0 +3 0 *
So

0100000
+2 0 *

+1 0
"""


def parabolic_dish(f=2.5, h=4.0, thickness=1.5):
  """a parabolic MIRROR blank: the cylinder rho <= r, -thickness <= z <= h minus the paraboloid's inside -- the
  paraboloid face bounds the material from OUTSIDE (a Cut by the paraboloid)"""
  f, h = float(f), float(h)
  r = float(2.0 * np.sqrt(f * h))
  tp = float(2.0 * np.pi)
  z0 = -float(thickness)
  return f"""CASCADE Topology V1, (c) Matra-Datavision
Locations 0
Curve2ds 9
1 0 0 0 1 
1 {tp!r} 0 0 1 
1 0 {r!r} 1 0 
1 0 0 1 0 
1 0 0 0 1 
1 {tp!r} 0 0 1 
1 0 {h - z0!r} 1 0 
1 0 0 1 0 
2 0 0 1 0 -0 1 {r!r}
Curves 4
4 0 0 0 0 1 0 0 0 1 1 0 0 {f!r}
2 0 0 {h!r} 0 0 1 1 0 -0 -0 1 0 {r!r}
1 {r!r} 0 {z0!r} 0 0 1 
2 0 0 {z0!r} 0 0 1 1 0 -0 -0 1 0 {r!r}
Polygon3D 0
PolygonOnTriangulations 0
Surfaces 3
7 0 0 0 0 0 1 
4 0 0 0 0 1 0 0 0 1 1 0 0 {f!r}
2 0 0 {z0!r} 0 0 1 1 0 -0 -0 1 0 {r!r}
1 0 0 {z0!r} 0 0 1 1 0 -0 -0 1 0 
Triangulations 0

TShapes 16
Ve
1e-07
0 0 0
0 0

0101101
*
Ve
1e-07
{r!r} 0 {h!r}
0 0

0101101
*
Ve
1e-07
{r!r} 0 {z0!r}
0 0

0101101
*
Ed
 1e-07 1 1 0
1  1 0 0 {r!r}
3  1 2CN 1 0 0 {r!r}
0

0101000
+16 0 -15 0 *
Ed
 1e-07 1 1 0
1  2 0 0 {tp!r}
2  3 1 0 0 {tp!r}
2  7 2 0 0 {tp!r}
0

0101000
+15 0 -15 0 *
Ed
 1e-07 1 1 1
2  4 1 0 0 {tp!r}
0

0101000
+16 0 -16 0 *
Ed
 1e-07 1 1 0
1  3 0 0 {h - z0!r}
3  5 6CN 2 0 0 {h - z0!r}
0

0101000
+14 0 -15 0 *
Ed
 1e-07 1 1 0
1  4 0 0 {tp!r}
2  8 2 0 0 {tp!r}
2  9 3 0 0 {tp!r}
0

0101000
+14 0 -14 0 *
Wi

0101100
+13 0 +12 0 -13 0 -11 0 *
Fa
0  1e-07 1 0

0101000
+8 0 *
Wi

0101100
+10 0 +12 0 -10 0 -9 0 *
Fa
0  1e-07 2 0

0101000
+6 0 *
Wi

0101100
+9 0 *
Fa
0  1e-07 3 0

0101000
+4 0 *
Sh

0101100
-7 0 +5 0 -3 0 *
So

0100000
+2 0 *

+1 0
"""
