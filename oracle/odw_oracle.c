/*
 * odw_oracle.c -- CPU restatement of the reference's Monte-Carlo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link
 * or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.  Plain C, float64, scalar, one ray at a time, brute
 * force over every face -- written for clarity, not speed.
 *
 * Parity status: the SAMPLER part (interp, row selection, draw order) is
 * pinned bit-for-bit against numpy.interp and against outputs of the
 * reference's own random_number_generator.py (tests/golden/sampler_*.npz).
 * The TRACING part is "parity unpinned" at the per-ray level: the reference
 * delegates intersections to FreeCAD/OpenCASCADE, which exists neither here
 * nor on the GPU box, and ships no known-answer vectors for intersections.
 * It follows the source text of the files cited below and is pinned only by
 * closed-form physics checks and the reference's statistical acceptance
 * tests (tests/test_oracle_physics.py).
 *
 * Reference files followed (relative to freecad/optics_design_workbench/):
 *   freecad_elements/ray.py:36-281      traceRay bounce loop / state machine
 *   freecad_elements/ray.py:290-452     findNearestIntersection
 *   freecad_elements/ray.py:455-495     getNormal, mirror, snellsLaw
 *   freecad_elements/ray.py:497-539     lineGrating
 *   freecad_elements/point_source.py:411-460   _makeRay
 *   distributions/random_number_generator.py:413-456, 467-560  draw/interp
 *   freecad_elements/find.py:79-104     relevantOpticalObjects
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, optional OpenMP).
 */
#include "../include/odw_trace.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11), counter-based RNG.            */
/* The reference seeds MT19937 from pid*time (simulation_loop.py:813), */
/* i.e. it is non-reproducible; a counter RNG keyed by the global ray  */
/* index replaces it so that results do not depend on sharding.        */
/* ------------------------------------------------------------------ */
void odw_oracle_philox(const uint32_t ctr_in[4], const uint32_t key_in[2],
                       uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; ++r) {
    if (r > 0) {
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in [0,1) from two 32-bit words, the construction numpy's
 * legacy random_sample uses (a>>5, b>>6).                                  */
static double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* Uniforms of one ray: counter = (ray_lo, ray_hi, slot, 0), key = seed.
 * slot 0: words 0,1 -> u_phi ; words 2,3 -> u_theta.
 * The reference draws u_phi first, then u_theta
 * (random_number_generator.py:492-498, reversed variable order).           */
static void ray_uniforms(uint64_t ray, uint64_t seed, double* u_phi, double* u_t) {
  uint32_t ctr[4] = {(uint32_t)ray, (uint32_t)(ray >> 32), 0u, 0u};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4];
  odw_oracle_philox(ctr, key, w);
  *u_phi = u53(w[0], w[1]);
  *u_t = u53(w[2], w[3]);
}

/* ------------------------------------------------------------------ */
/* numpy.interp restated (numpy/_core/src/multiarray/compiled_base.c,  */
/* arr_interp).  Used by the reference at                              */
/* random_number_generator.py:445.                                     */
/* ------------------------------------------------------------------ */
double odw_oracle_interp(double x, const double* xp, const double* fp, int32_t n) {
  if (isnan(x)) return x;
  if (x < xp[0]) return fp[0];
  if (x > xp[n - 1]) return fp[n - 1];
  /* j = last index with xp[j] <= x */
  int32_t lo = 0, hi = n; /* invariant: xp[lo] <= x, (hi==n or xp[hi] > x) */
  while (hi - lo > 1) {
    int32_t mid = lo + (hi - lo) / 2;
    if (x >= xp[mid]) lo = mid; else hi = mid;
  }
  int32_t j = lo;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (isnan(r)) {
    r = slope * (x - xp[j + 1]) + fp[j + 1];
    if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

/* VectorRandomVariable.draw for one sample given its two uniforms
 * (random_number_generator.py:413-456): phi from the marginal table, then
 * the conditional row nearest to phi (argmin over mid-points, first minimum),
 * then theta (or r) from that row.                                         */
static void sample_one(const odw_source_desc* s, double u_phi, double u_t,
                       double* t_out, double* phi_out) {
  double phi = odw_oracle_interp(u_phi, s->phi_cdf, s->phi_edges, s->n_phi_knots);
  int32_t row = 0;
  if (s->n_t_rows > 1) {
    double best = INFINITY;
    for (int32_t i = 0; i < s->n_t_rows; ++i) {
      double mid = (s->phi_edges[i + 1] + s->phi_edges[i]) / 2;
      double d = fabs(mid - phi);
      if (d < best) { best = d; row = i; }
    }
  }
  const double* cdf = s->t_cdf + (size_t)row * (size_t)s->n_t_knots;
  *t_out = odw_oracle_interp(u_t, cdf, s->t_edges, s->n_t_knots);
  *phi_out = phi;
}

int odw_oracle_sample(const odw_source_desc* s, uint64_t first, uint64_t n,
                      uint64_t seed, double* t_out, double* phi_out) {
  for (uint64_t i = 0; i < n; ++i) {
    double up, ut;
    ray_uniforms(first + i, seed, &up, &ut);
    sample_one(s, up, ut, &t_out[i], &phi_out[i]);
  }
  return ODW_OK;
}

/* the same with caller-supplied uniforms (pins against the reference's
 * numpy-seeded draws)                                                      */
int odw_oracle_sample_uniforms(const odw_source_desc* s, uint64_t n,
                               const double* u_phi, const double* u_t,
                               double* t_out, double* phi_out) {
  for (uint64_t i = 0; i < n; ++i) sample_one(s, u_phi[i], u_t[i], &t_out[i], &phi_out[i]);
  return ODW_OK;
}

/* ------------------------------------------------------------------ */
/* small vector helpers                                                */
/* ------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;
static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 mul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static double len(v3 a) { return sqrt(dot(a, a)); }
/* point transform with rows (R|t) */
static v3 xf_point(const double* m, v3 p) {
  return V(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
           m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
           m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
/* inverse of a rigid (R|t): p = R^T (q - t) */
static v3 xf_point_inv(const double* m, v3 q) {
  v3 d = V(q.x - m[3], q.y - m[7], q.z - m[11]);
  return V(m[0] * d.x + m[4] * d.y + m[8] * d.z,
           m[1] * d.x + m[5] * d.y + m[9] * d.z,
           m[2] * d.x + m[6] * d.y + m[10] * d.z);
}

/* ------------------------------------------------------------------ */
/* PointSourceProxy._makeRay (point_source.py:411-460)                 */
/* ------------------------------------------------------------------ */
static void make_ray(const odw_source_desc* s, double t_or_r, double phi,
                     v3* origin, v3* dir) {
  v3 ldir, lorg;
  if (isfinite(s->focal_length)) {
    double theta = t_or_r;
    /* Rz(phi) * Rx(theta) * (0,0,1) */
    double st = sin(theta), ct = cos(theta), sp = sin(phi), cp = cos(phi);
    ldir = V(st * sp, -st * cp, ct);
    lorg = mul(sub(V(0, 0, 1), ldir), s->focal_length);
  } else {
    double r = t_or_r;
    ldir = V(0, 0, 1);
    /* r*x^*cos(phi) + r*(x^ x z^)*sin(phi), x^ x z^ = (0,-1,0) */
    lorg = add(mul(V(1, 0, 0), r * cos(phi)), mul(V(0, -1, 0), r * sin(phi)));
  }
  v3 p1 = lorg;
  v3 p2 = add(lorg, mul(ldir, 1.0 / len(ldir)));
  p1 = xf_point(s->xform, p1);
  p2 = xf_point(s->xform, p2);
  v3 d = sub(p2, p1);
  *origin = p1;
  *dir = mul(d, 1.0 / len(d));
}

/* ------------------------------------------------------------------ */
/* polynomial root isolation for the torus quartic                     */
/* ------------------------------------------------------------------ */
static double poly(const double* c, int deg, double t) {
  double r = c[deg];
  for (int i = deg - 1; i >= 0; --i) r = r * t + c[i];
  return r;
}

/* root of a function monotone on [a,b] with f(a)*f(b) < 0: bisection safeguarded Newton */
static double mono_root(const double* c, int deg, double a, double b) {
  double dc[5];
  for (int i = 1; i <= deg; ++i) dc[i - 1] = c[i] * i;
  double fa = poly(c, deg, a);
  double lo = a, hi = b;
  if (fa > 0) { lo = b; hi = a; } /* f(lo) < 0 < f(hi) */
  double x = 0.5 * (a + b);
  for (int it = 0; it < 200; ++it) {
    double f = poly(c, deg, x);
    if (f == 0) return x;
    if (f < 0) lo = x; else hi = x;
    double df = poly(dc, deg - 1, x);
    double xn = (df != 0) ? x - f / df : 0.5 * (lo + hi);
    double mn = lo < hi ? lo : hi, mx = lo < hi ? hi : lo;
    if (!(xn > mn && xn < mx)) xn = 0.5 * (lo + hi);
    if (xn == x || fabs(hi - lo) <= 4e-16 * (fabs(lo) + fabs(hi))) return xn;
    x = xn;
  }
  return x;
}

/* all sign-change roots of a polynomial of degree <= 4 in [lo,hi], ascending */
static int poly_roots(const double* c, int deg, double lo, double hi, double* out) {
  while (deg > 0 && c[deg] == 0) --deg;
  if (deg <= 0) return 0;
  if (deg == 1) {
    double r = -c[0] / c[1];
    if (r >= lo && r <= hi) { out[0] = r; return 1; }
    return 0;
  }
  double dc[5], crit[4];
  for (int i = 1; i <= deg; ++i) dc[i - 1] = c[i] * i;
  int nc = poly_roots(dc, deg - 1, lo, hi, crit);
  double brk[6];
  int nb = 0;
  brk[nb++] = lo;
  for (int i = 0; i < nc; ++i) brk[nb++] = crit[i];
  brk[nb++] = hi;
  int n = 0;
  for (int i = 0; i + 1 < nb; ++i) {
    double a = brk[i], b = brk[i + 1];
    if (!(b > a)) continue;
    double fa = poly(c, deg, a), fb = poly(c, deg, b);
    if (fa == 0) {
      if (n == 0 || out[n - 1] != a) out[n++] = a;
      continue;
    }
    if ((fa < 0 && fb > 0) || (fa > 0 && fb < 0)) out[n++] = mono_root(c, deg, a, b);
    else if (fb == 0 && i + 2 == nb) out[n++] = b;
  }
  return n;
}

/* ------------------------------------------------------------------ */
/* primitives: candidate intersections and inside tests                */
/* ------------------------------------------------------------------ */
typedef struct {
  double t;     /* distance along the (unit) ray                     */
  v3 n_local;   /* outward face normal in the primitive's frame      */
  int face;
} cand;

static int quad_roots(double a, double b_half, double c, double* r) {
  /* a t^2 + 2 b_half t + c = 0 */
  if (a == 0) {
    if (b_half == 0) return 0;
    r[0] = -c / (2 * b_half);
    return 1;
  }
  double disc = b_half * b_half - a * c;
  if (disc < 0) return 0;
  double sq = sqrt(disc);
  double q = -(b_half + (b_half >= 0 ? sq : -sq));
  double t0 = q / a;
  double t1 = (q != 0) ? c / q : t0;
  if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
  r[0] = t0; r[1] = t1;
  return 2;
}

/* signed-distance-like value of local point p w.r.t. primitive (negative inside) */
static double prim_sdist(int type, const double* par, v3 p) {
  switch (type) {
    case ODW_PRIM_BOX: {
      double dx = fmax(-p.x, p.x - par[0]);
      double dy = fmax(-p.y, p.y - par[1]);
      double dz = fmax(-p.z, p.z - par[2]);
      return fmax(dx, fmax(dy, dz));
    }
    case ODW_PRIM_SPHERE: return len(p) - par[0];
    case ODW_PRIM_CYLINDER: {
      double rho = sqrt(p.x * p.x + p.y * p.y);
      return fmax(rho - par[0], fmax(-p.z, p.z - par[1]));
    }
    case ODW_PRIM_CONE: {
      double k = (par[1] - par[0]) / par[2];
      double rho = sqrt(p.x * p.x + p.y * p.y);
      double lat = (rho - (par[0] + k * p.z)) / sqrt(1 + k * k);
      return fmax(lat, fmax(-p.z, p.z - par[2]));
    }
    case ODW_PRIM_TORUS: {
      double rho = sqrt(p.x * p.x + p.y * p.y);
      double a = rho - par[0];
      return sqrt(a * a + p.z * p.z) - par[1];
    }
    case ODW_PRIM_PARABOLOID: {
      /* x^2 + y^2 - 4 f z over the length of its gradient: the distance to first order */
      double r2 = p.x * p.x + p.y * p.y;
      double lat = (r2 - 4.0 * par[0] * p.z) / (2.0 * sqrt(r2 + 4.0 * par[0] * par[0]));
      return fmax(lat, p.z - par[1]);
    }
  }
  return INFINITY;
}

/* enumerate every intersection of the unit ray (o,d) (local frame) with the
 * UNTRIMMED face surfaces of one primitive, then apply the natural face trim
 * with tolerance tol -- the analogue of line.Curve.intersect(surface) +
 * vert.distToShape(face) < distTol (ray.py:411-426).                       */
static int prim_candidates(int type, const double* par, int facemask, v3 o, v3 d,
                           double tol, cand* out) {
  int n = 0;
  const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  switch (type) {
    case ODW_PRIM_BOX: {
      for (int f = 0; f < 6; ++f) {
        if (!((facemask >> f) & 1)) continue;
        int a = f >> 1;
        if (dd[a] == 0) continue;
        double c = (f & 1) ? par[a] : 0.0;
        double t = (c - oo[a]) / dd[a];
        int b1 = (a + 1) % 3, b2 = (a + 2) % 3;
        double p1 = oo[b1] + t * dd[b1], p2 = oo[b2] + t * dd[b2];
        if (p1 < -tol || p1 > par[b1] + tol || p2 < -tol || p2 > par[b2] + tol) continue;
        double nn[3] = {0, 0, 0};
        nn[a] = (f & 1) ? 1.0 : -1.0;
        out[n].t = t; out[n].n_local = V(nn[0], nn[1], nn[2]); out[n].face = f; ++n;
      }
      break;
    }
    case ODW_PRIM_SPHERE: {
      if (!(facemask & 1)) break;
      double r[2];
      int nr = quad_roots(dot(d, d), dot(o, d), dot(o, o) - par[0] * par[0], r);
      for (int i = 0; i < nr; ++i) {
        v3 p = add(o, mul(d, r[i]));
        out[n].t = r[i]; out[n].n_local = mul(p, 1.0 / len(p)); out[n].face = 0; ++n;
      }
      break;
    }
    case ODW_PRIM_CYLINDER:
    case ODW_PRIM_CONE: {
      double R1 = par[0], R2, H, k;
      if (type == ODW_PRIM_CYLINDER) { R2 = par[0]; H = par[1]; k = 0; }
      else { R2 = par[1]; H = par[2]; k = (R2 - R1) / H; }
      if (facemask & 1) {
        double r[2];
        double rz = R1 + k * o.z;
        double a = d.x * d.x + d.y * d.y - k * k * d.z * d.z;
        double bh = o.x * d.x + o.y * d.y - k * rz * d.z;
        double c = o.x * o.x + o.y * o.y - rz * rz;
        int nr = quad_roots(a, bh, c, r);
        for (int i = 0; i < nr; ++i) {
          v3 p = add(o, mul(d, r[i]));
          if (p.z < -tol || p.z > H + tol) continue;
          double rr = R1 + k * p.z;
          if (rr < -tol) continue; /* other nappe of the cone */
          v3 g = V(p.x, p.y, -k * rr);
          double gl = len(g);
          if (gl == 0) continue;
          out[n].t = r[i]; out[n].n_local = mul(g, 1.0 / gl); out[n].face = 0; ++n;
        }
      }
      for (int f = 1; f <= 2; ++f) {
        if (!((facemask >> f) & 1)) continue;
        if (d.z == 0) continue;
        double zc = (f == 1) ? 0.0 : H;
        double rc = (f == 1) ? R1 : R2;
        if (rc <= 0) continue;
        double t = (zc - o.z) / d.z;
        double px = o.x + t * d.x, py = o.y + t * d.y;
        if (px * px + py * py > (rc + tol) * (rc + tol)) continue;
        out[n].t = t; out[n].n_local = V(0, 0, f == 1 ? -1.0 : 1.0); out[n].face = f; ++n;
      }
      break;
    }
    case ODW_PRIM_PARABOLOID: {
      /* the solid x^2 + y^2 <= 4 f z, z <= H: lateral face 0, cap z = H face 2 (rim radius 2 sqrt(f H)) */
      double f = par[0], H = par[1];
      if (facemask & 1) {
        double r[2];
        int nr = quad_roots(d.x * d.x + d.y * d.y, o.x * d.x + o.y * d.y - 2.0 * f * d.z,
                            o.x * o.x + o.y * o.y - 4.0 * f * o.z, r);
        for (int i = 0; i < nr; ++i) {
          v3 p = add(o, mul(d, r[i]));
          if (p.z < -tol || p.z > H + tol) continue;
          v3 g = V(p.x, p.y, -2.0 * f);
          out[n].t = r[i]; out[n].n_local = mul(g, 1.0 / len(g)); out[n].face = 0; ++n;
        }
      }
      if ((facemask & 4) && d.z != 0) {
        double rc = 2.0 * sqrt(f * H);
        double t = (H - o.z) / d.z;
        double px = o.x + t * d.x, py = o.y + t * d.y;
        if (px * px + py * py <= (rc + tol) * (rc + tol)) {
          out[n].t = t; out[n].n_local = V(0, 0, 1.0); out[n].face = 2; ++n;
        }
      }
      break;
    }
    case ODW_PRIM_TORUS: {
      if (!(facemask & 1)) break;
      double R1 = par[0], R2 = par[1];
      /* re-origin at the closest approach to the torus centre */
      double dl = dot(d, d);
      double t0 = -dot(o, d) / dl;
      v3 q = add(o, mul(d, t0));
      double A = dl;
      double C = dot(q, q) + R1 * R1 - R2 * R2;
      double E = d.x * d.x + d.y * d.y;
      double F = 2 * (q.x * d.x + q.y * d.y);
      double G = q.x * q.x + q.y * q.y;
      double B = 2 * dot(q, d); /* ~0 by construction, kept for exactness */
      double c[5];
      c[4] = A * A;
      c[3] = 2 * A * B;
      c[2] = B * B + 2 * A * C - 4 * R1 * R1 * E;
      c[1] = 2 * B * C - 4 * R1 * R1 * F;
      c[0] = C * C - 4 * R1 * R1 * G;
      double bound = (R1 + R2) * 1.0000001 + 1e-9;
      /* all roots lie inside the bounding sphere: |s| <= bound/sqrt(A) */
      double sb = bound / sqrt(A);
      double roots[4];
      int nr = poly_roots(c, 4, -sb, sb, roots);
      for (int i = 0; i < nr; ++i) {
        double t = roots[i] + t0;
        v3 p = add(o, mul(d, t));
        double rho = sqrt(p.x * p.x + p.y * p.y);
        if (rho == 0) continue;
        v3 g = V(p.x - R1 * p.x / rho, p.y - R1 * p.y / rho, p.z);
        double gl = len(g);
        if (gl == 0) continue;
        out[n].t = t; out[n].n_local = mul(g, 1.0 / gl); out[n].face = 0; ++n;
      }
      break;
    }
  }
  return n;
}

/* squared distance of the point w from the segment 0 -> e */
static double seg_dist2(v3 w, v3 e) {
  double s = dot(w, e) / dot(e, e);
  if (s < 0) s = 0;
  if (s > 1) s = 1;
  v3 r = sub(w, mul(e, s));
  return dot(r, r);
}

/* ------------------------------------------------------------------ */
/* findNearestIntersection (ray.py:290-452)                            */
/* ------------------------------------------------------------------ */
typedef struct {
  int found;
  int prim, face, group;
  double dist;
  v3 point;    /* global */
  v3 normal;   /* global outward normal of the solid at the hit (unit) */
} nearest_hit;

static uint64_t relevant_mask(const odw_scene_desc* sc, int seq_idx) {
  /* find.py:79-104 */
  uint64_t all = (sc->n_groups >= 64) ? ~0ull : ((1ull << sc->n_groups) - 1);
  uint64_t m = all;
  if (sc->seq_enabled) m = (seq_idx < sc->seq_len) ? sc->seq_mask[seq_idx] : 0ull;
  return m & ~sc->ignore_mask & all;
}

static int better(double t, int prim, int face, double bt, int bprim, int bface) {
  if (t != bt) return t < bt;
  if (prim != bprim) return prim < bprim;
  return face < bface;
}

static nearest_hit nearest_skipping(const odw_scene_desc* sc, const odw_limits* lim, v3 start,
                                    v3 dir, int medium, int seq_idx, int skip_solid);

static nearest_hit nearest(const odw_scene_desc* sc, const odw_limits* lim, v3 start,
                           v3 dir, int medium, int seq_idx) {
  return nearest_skipping(sc, lim, start, dir, medium, seq_idx, -1);
}

/* skip_solid: the convex solid the ray has just left (ODW_FLAG_CONVEX,
 * include/odw_trace.h): a straight line meets a convex solid in one interval,
 * so none of its faces can be met again; -1: none */
static nearest_hit nearest_skipping(const odw_scene_desc* sc, const odw_limits* lim, v3 start,
                                    v3 dir, int medium, int seq_idx, int skip_solid) {
  const double tol = lim->dist_tol;
  const double max_len = lim->max_ray_length;
  uint64_t mask = relevant_mask(sc, seq_idx);
  v3 dn = mul(dir, 1.0 / len(dir));

  nearest_hit any = {0}, oth = {0};
  any.dist = INFINITY; oth.dist = INFINITY;
  any.prim = oth.prim = any.face = oth.face = 0x7fffffff;

  for (int p = 0; p < sc->n_prims; ++p) {
    int g = sc->prim_group[p];
    if (!((mask >> g) & 1)) continue;
    if (skip_solid >= 0 && sc->prim_solid && sc->prim_solid[p] == skip_solid) continue;
    const double* M = sc->prim_xform + 12 * (size_t)p;
    if (sc->prim_type[p] == ODW_PRIM_TRIANGLE) {
      /* one facet of a tessellated face (include/odw_trace.h): v0, v1, v2 in
       * global coordinates; a hit within distTol of the facet counts */
      v3 v0 = V(M[0], M[1], M[2]);
      v3 e1 = sub(V(M[3], M[4], M[5]), v0), e2 = sub(V(M[6], M[7], M[8]), v0);
      v3 nn = cross(e1, e2);
      double a2 = len(nn);
      v3 pv = cross(dn, e2);
      double det = dot(e1, pv);
      if (det == 0) continue;
      v3 tv = sub(start, v0);
      double u = dot(tv, pv) / det;
      v3 qv = cross(tv, e1);
      double v = dot(dn, qv) / det;
      /* per edge: reach of the tolerance in barycentric units; edges shared with a neighbouring
       * facet of the same face (tri_edges) are closed up to rounding */
      int fe = sc->tri_edges ? sc->tri_edges[p] : 7;
      double a0 = (fe & 1) ? tol * (len(e2) / a2) : 1e-9;
      double a1 = (fe & 2) ? tol * (len(e1) / a2) : 1e-9;
      double a2e = (fe & 4) ? tol * (len(sub(e2, e1)) / a2) : 1e-9;
      if (u < -a0 || v < -a1 || u + v > 1.0 + a2e) continue;
      if ((u < 0 && (fe & 1)) || (v < 0 && (fe & 2)) || (u + v > 1.0 && (fe & 4))) {
        /* beyond an edge of the face, inside the parallelogram the bounds above allow (far too
         * long for slivers): the distance to the facet itself decides */
        v3 w = add(mul(e1, u), mul(e2, v));              /* hit point - v0, in the facet's plane */
        double best = seg_dist2(w, e1);
        double other = seg_dist2(w, e2);
        if (other < best) best = other;
        other = seg_dist2(sub(w, e1), sub(e2, e1));
        if (other < best) best = other;
        if (best > tol * tol) continue;
      }
      double t = dot(e2, qv) / det;
      if (!(t > tol) || !(t < max_len + tol)) continue;
      v3 gp = add(start, mul(dn, t));
      v3 ng = mul(nn, 1.0 / a2);
      if (sc->tri_normals) {
        const double* vn = sc->tri_normals + 9 * (size_t)p;
        v3 tg = sub(gp, v0);
        double inv = 1.0 / dot(nn, nn);
        double bu = dot(cross(tg, e2), nn) * inv, bv = dot(cross(e1, tg), nn) * inv, bw = 1.0 - bu - bv;
        v3 ni = V(bw * vn[0] + bu * vn[3] + bv * vn[6], bw * vn[1] + bu * vn[4] + bv * vn[7],
                  bw * vn[2] + bu * vn[5] + bv * vn[8]);
        if (dot(ni, ni) > 0) ng = mul(ni, 1.0 / len(ni));
      }
      if (sc->prim_flags[p] & ODW_FLAG_FLIP_NORMAL) ng = mul(ng, -1.0);
      nearest_hit h;
      h.found = 1; h.prim = p; h.face = 0; h.group = g; h.dist = t;
      h.point = gp; h.normal = ng;
      if (better(t, p, 0, any.dist, any.prim, any.face)) any = h;
      if (g != medium && better(t, p, 0, oth.dist, oth.prim, oth.face)) oth = h;
      continue;
    }
    /* ray in local coordinates: lstart = M*start, ldir = M*(start+dir)-lstart
     * (ray.py:348-349) */
    v3 lstart = xf_point(M, start);
    v3 ldir = sub(xf_point(M, add(start, dn)), lstart);
    int flags = sc->prim_flags[p];
    cand cs[8];
    int nc = prim_candidates(sc->prim_type[p], sc->prim_params + 4 * (size_t)p,
                             flags >> ODW_FACEMASK_SHIFT, lstart, ldir, tol, cs);
    for (int i = 0; i < nc; ++i) {
      double t = cs[i].t;
      v3 lp = add(lstart, mul(ldir, t));
      double dist = len(sub(lp, lstart));
      /* (vec-lstart).Length > distTol ; on the finite line of length
       * maxRayLength within distTol (ray.py:424-425) */
      if (!(t > 0) || !(dist > tol) || !(dist < max_len + tol)) continue;
      /* trimmed face: CSG conditions (boolean results keep only the part of a
       * face inside/outside the other operands) */
      v3 gp = xf_point_inv(M, lp);
      int ok = 1;
      for (int c = sc->prim_cond_off[p]; c < sc->prim_cond_off[p + 1] && ok; ++c) {
        int q = sc->cond_prim[c];
        v3 qp = xf_point(sc->prim_xform + 12 * (size_t)q, gp);
        double sd = prim_sdist(sc->prim_type[q], sc->prim_params + 4 * (size_t)q, qp);
        if (sc->cond_inside[c]) { if (sd > tol) ok = 0; }
        else { if (sd < -tol) ok = 0; }
      }
      if (!ok) continue;
      v3 nl = cs[i].n_local;
      if (flags & ODW_FLAG_FLIP_NORMAL) nl = mul(nl, -1.0);
      /* normal to global: gpM*(p+n) - p_global (ray.py:476) */
      v3 ng = sub(xf_point_inv(M, add(lp, nl)), gp);
      nearest_hit h;
      h.found = 1; h.prim = p; h.face = cs[i].face; h.group = g; h.dist = dist;
      h.point = gp; h.normal = ng;
      if (better(dist, p, h.face, any.dist, any.prim, any.face)) any = h;
      if (g != medium && better(dist, p, h.face, oth.dist, oth.prim, oth.face)) oth = h;
    }
  }
  if (!any.found) return any;
  /* keep hits closer than min+2*distTol, prefer one whose group is not the
   * current medium (ray.py:438-452) */
  if (oth.found && oth.dist < any.dist + 2 * tol) return oth;
  return any;
}

/* ------------------------------------------------------------------ */
/* REFERENCE-STRICT findNearestIntersection (ray.py:328-452).          */
/*                                                                     */
/* nearest_skipping() above is the rule the device implements: order-  */
/* independent (two running minima) and with the convex-solid skip.    */
/* This second restatement keeps the reference's own control flow so   */
/* that the two can be compared ray for ray (tests/test_oracle_strict) */
/*   - no convex-solid skip: every shell of every relevant group is a  */
/*     candidate of every segment (ray.py:328-364);                    */
/*   - shells sorted by the distance of their (distTol-enlarged)       */
/*     bounding box from the start (stable, ray.py:367), skipped when  */
/*     that distance is no longer < maxRayLength or the line misses    */
/*     the box (ray.py:372-374);                                       */
/*   - faces of a shell sorted the same way (ray.py:383-404), each     */
/*     intersected as an infinite line with the untrimmed surface      */
/*     (ray.py:411); a point is kept iff it is further than distTol    */
/*     from the start, within distTol of the FINITE line of the length */
/*     maxRayLength had when the shell was entered (ray.py:377,425)    */
/*     and within distTol of the trimmed face (ray.py:426);            */
/*   - every kept point shrinks maxRayLength to its distance +         */
/*     5*distTol (ray.py:432);                                         */
/*   - selection: keep points closer than min + 2*distTol, stable sort */
/*     by distance, first one whose group is not the current medium,   */
/*     else the first (ray.py:438-452).                                */
/* What it cannot restate (no OpenCASCADE): bounding boxes are those   */
/* of the primitives in GLOBAL axes (the reference's are the shell's   */
/* and face's in the group's local axes) -- boxes only order and cull  */
/* candidates, they never decide whether a point is valid --, and      */
/* "within distTol of the trimmed face" stays the conjunction of       */
/* signed-distance tests of the default mode (DESIGN.md section 3,     */
/* note 1).                                                            */
/* ------------------------------------------------------------------ */
static int g_strict = 0;
int odw_oracle_set_strict(int on) { g_strict = on != 0; return ODW_OK; }
int odw_oracle_get_strict(void) { return g_strict; }

typedef struct { double lo[3], hi[3]; } aabb;

static void aabb_empty(aabb* b) {
  for (int a = 0; a < 3; ++a) { b->lo[a] = INFINITY; b->hi[a] = -INFINITY; }
}
static void aabb_add_point(aabb* b, v3 p) {
  const double c[3] = {p.x, p.y, p.z};
  for (int a = 0; a < 3; ++a) { if (c[a] < b->lo[a]) b->lo[a] = c[a]; if (c[a] > b->hi[a]) b->hi[a] = c[a]; }
}
static void aabb_merge(aabb* b, const aabb* o) {
  for (int a = 0; a < 3; ++a) { if (o->lo[a] < b->lo[a]) b->lo[a] = o->lo[a]; if (o->hi[a] > b->hi[a]) b->hi[a] = o->hi[a]; }
}
/* BoundBox.isInside / closestPoint distance (ray.py:353-357): 0 inside */
static double aabb_dist(const aabb* b, v3 p) {
  const double c[3] = {p.x, p.y, p.z};
  double s = 0;
  for (int a = 0; a < 3; ++a) {
    double d = 0;
    if (c[a] < b->lo[a]) d = b->lo[a] - c[a]; else if (c[a] > b->hi[a]) d = c[a] - b->hi[a];
    s += d * d;
  }
  return sqrt(s);
}
/* BoundBox.intersect(base, dir): the infinite line meets the box (slab test) */
static int aabb_line(const aabb* b, v3 o, v3 d) {
  const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  double t0 = -INFINITY, t1 = INFINITY;
  for (int a = 0; a < 3; ++a) {
    if (dd[a] == 0) { if (oo[a] < b->lo[a] || oo[a] > b->hi[a]) return 0; continue; }
    double ta = (b->lo[a] - oo[a]) / dd[a], tb = (b->hi[a] - oo[a]) / dd[a];
    if (ta > tb) { double t = ta; ta = tb; tb = t; }
    if (ta > t0) t0 = ta;
    if (tb < t1) t1 = tb;
  }
  return t0 <= t1;
}

/* box of one primitive in global axes, enlarged by tol (cachedBoundBox(..., enlarge=distTol)) */
static void prim_aabb(const odw_scene_desc* sc, int p, double tol, aabb* out) {
  const double* M = sc->prim_xform + 12 * (size_t)p;
  const double* par = sc->prim_params + 4 * (size_t)p;
  aabb_empty(out);
  if (sc->prim_type[p] == ODW_PRIM_TRIANGLE) {
    for (int k = 0; k < 3; ++k) aabb_add_point(out, V(M[3 * k], M[3 * k + 1], M[3 * k + 2]));
  } else {
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    switch (sc->prim_type[p]) {
      case ODW_PRIM_BOX: hi[0] = par[0]; hi[1] = par[1]; hi[2] = par[2]; break;
      case ODW_PRIM_SPHERE: for (int a = 0; a < 3; ++a) { lo[a] = -par[0]; hi[a] = par[0]; } break;
      case ODW_PRIM_CYLINDER: lo[0] = lo[1] = -par[0]; hi[0] = hi[1] = par[0]; hi[2] = par[1]; break;
      case ODW_PRIM_CONE: { double r = fmax(par[0], par[1]); lo[0] = lo[1] = -r; hi[0] = hi[1] = r; hi[2] = par[2]; break; }
      case ODW_PRIM_TORUS: { double r = par[0] + par[1]; lo[0] = lo[1] = -r; hi[0] = hi[1] = r; lo[2] = -par[1]; hi[2] = par[1]; break; }
      case ODW_PRIM_PARABOLOID: { double r = 2.0 * sqrt(par[0] * par[1]); lo[0] = lo[1] = -r; hi[0] = hi[1] = r; hi[2] = par[1]; break; }
    }
    for (int k = 0; k < 8; ++k)
      aabb_add_point(out, xf_point_inv(M, V((k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2])));
  }
  for (int a = 0; a < 3; ++a) { out->lo[a] -= tol; out->hi[a] += tol; }
}

/* the static part of the candidate search, built once per scene (raytracing_cache.py:92-111):
 * shells in group order then shell order, their faces' primitives, all boxes */
typedef struct { int group, solid, first, count; aabb box; } shell_rec;
typedef struct {
  const odw_scene_desc* sc; const double* xform; int n_prims; double tol;
  int n_shells; shell_rec* shells; int* prims; aabb* prim_box;
} strict_cache;
static strict_cache g_sc = {0};

static void strict_cache_build(const odw_scene_desc* sc, double tol) {
  free(g_sc.shells); free(g_sc.prims); free(g_sc.prim_box);
  memset(&g_sc, 0, sizeof g_sc);
  g_sc.sc = sc; g_sc.xform = sc->prim_xform; g_sc.n_prims = sc->n_prims; g_sc.tol = tol;
  g_sc.prim_box = (aabb*)malloc((size_t)(sc->n_prims ? sc->n_prims : 1) * sizeof(aabb));
  g_sc.prims = (int*)malloc((size_t)(sc->n_prims ? sc->n_prims : 1) * sizeof(int));
  g_sc.shells = (shell_rec*)malloc((size_t)(sc->n_prims ? sc->n_prims : 1) * sizeof(shell_rec));
  int* shell_of = (int*)malloc((size_t)(sc->n_prims ? sc->n_prims : 1) * sizeof(int));
  int n_sh = 0;
  for (int g = 0; g < sc->n_groups; ++g) {
    int first_of_group = n_sh;
    for (int p = 0; p < sc->n_prims; ++p) {
      if (sc->prim_group[p] != g) continue;
      shell_of[p] = -1;
      int facemask = sc->prim_type[p] == ODW_PRIM_TRIANGLE ? 1 : (sc->prim_flags[p] >> ODW_FACEMASK_SHIFT) & 0xff;
      if (!facemask) continue;                       /* an operand without a face of its own */
      prim_aabb(sc, p, tol, &g_sc.prim_box[p]);
      int solid = sc->prim_solid ? sc->prim_solid[p] : p;
      int k;
      for (k = n_sh - 1; k >= first_of_group; --k) if (g_sc.shells[k].solid == solid) break;
      if (k < first_of_group) {
        k = n_sh++;
        g_sc.shells[k].group = g; g_sc.shells[k].solid = solid; g_sc.shells[k].count = 0;
        g_sc.shells[k].box = g_sc.prim_box[p];
      } else {
        aabb_merge(&g_sc.shells[k].box, &g_sc.prim_box[p]);
      }
      shell_of[p] = k;
      g_sc.shells[k].count++;
    }
  }
  int at = 0;
  for (int k = 0; k < n_sh; ++k) { g_sc.shells[k].first = at; at += g_sc.shells[k].count; g_sc.shells[k].count = 0; }
  for (int g = 0; g < sc->n_groups; ++g)
    for (int p = 0; p < sc->n_prims; ++p)
      if (sc->prim_group[p] == g && shell_of[p] >= 0) {
        shell_rec* r = &g_sc.shells[shell_of[p]];
        g_sc.prims[r->first + r->count++] = p;
      }
  g_sc.n_shells = n_sh;
  free(shell_of);
}

/* called once per trace call, outside the parallel region */
static void strict_prepare(const odw_scene_desc* sc, double tol) {
  strict_cache_build(sc, tol);
}

typedef struct { int shell; double bb_dist; } shell_cand;
typedef struct { int prim, face; double bb_dist; } face_cand;

static int n_faces_of(int type) {
  switch (type) {
    case ODW_PRIM_BOX: return 6;
    case ODW_PRIM_CYLINDER: case ODW_PRIM_CONE: case ODW_PRIM_PARABOLOID: return 3;
    default: return 1;
  }
}

static nearest_hit nearest_strict(const odw_scene_desc* sc, const odw_limits* lim, v3 start,
                                  v3 dir, int medium, int seq_idx) {
  const double tol = lim->dist_tol;
  double max_len = lim->max_ray_length;
  uint64_t mask = relevant_mask(sc, seq_idx);
  v3 dn = mul(dir, 1.0 / len(dir));
  nearest_hit none;
  memset(&none, 0, sizeof none);
  none.dist = INFINITY;
  /* (the shell tables were built by strict_prepare() before any thread got here) */

  /* (1) shells of the relevant groups, in group order then shell order (ray.py:328-364) */
  shell_cand* sh = (shell_cand*)malloc((size_t)(g_sc.n_shells ? g_sc.n_shells : 1) * sizeof(shell_cand));
  int n_sh = 0;
  for (int k = 0; k < g_sc.n_shells; ++k) {
    if (!((mask >> g_sc.shells[k].group) & 1)) continue;
    double bd = aabb_dist(&g_sc.shells[k].box, start);
    if (!isfinite(max_len) || bd < max_len) { sh[n_sh].shell = k; sh[n_sh].bb_dist = bd; ++n_sh; }
  }
  /* (2) stable sort by bounding-box distance (ray.py:367) */
  for (int i = 1; i < n_sh; ++i) {
    shell_cand c = sh[i];
    int j = i - 1;
    while (j >= 0 && sh[j].bb_dist > c.bb_dist) { sh[j + 1] = sh[j]; --j; }
    sh[j + 1] = c;
  }

  int cap_hit = 16, n_hit = 0;
  nearest_hit* hits = (nearest_hit*)malloc((size_t)cap_hit * sizeof(nearest_hit));
  int cap_fc = 64;
  face_cand* fc = (face_cand*)malloc((size_t)cap_fc * sizeof(face_cand));

  for (int k = 0; k < n_sh; ++k) {
    const shell_rec* S = &g_sc.shells[sh[k].shell];
    /* (3) ray.py:372-374 */
    if (!(sh[k].bb_dist < max_len) || !aabb_line(&S->box, start, dn)) continue;
    const double line_len = max_len;   /* Part.makeLine(lstart, lstart + dir * maxRayLength), ray.py:377 */
    int n_fc = 0;
    for (int m = 0; m < S->count; ++m) {
      int p = g_sc.prims[S->first + m];
      int facemask = sc->prim_type[p] == ODW_PRIM_TRIANGLE ? 1 : (sc->prim_flags[p] >> ODW_FACEMASK_SHIFT) & 0xff;
      const aabb* pb = &g_sc.prim_box[p];
      double fd = aabb_dist(pb, start);
      if (!(fd < max_len) || !aabb_line(pb, start, dn)) continue;        /* ray.py:397-398 */
      for (int f = 0; f < n_faces_of(sc->prim_type[p]); ++f) {
        if (!((facemask >> f) & 1)) continue;
        if (n_fc == cap_fc) { cap_fc *= 2; fc = (face_cand*)realloc(fc, (size_t)cap_fc * sizeof(face_cand)); }
        fc[n_fc].prim = p; fc[n_fc].face = f; fc[n_fc].bb_dist = fd; ++n_fc;
      }
    }
    for (int i = 1; i < n_fc; ++i) {                                      /* ray.py:404 */
      face_cand c = fc[i];
      int j = i - 1;
      while (j >= 0 && fc[j].bb_dist > c.bb_dist) { fc[j + 1] = fc[j]; --j; }
      fc[j + 1] = c;
    }
    for (int i = 0; i < n_fc; ++i) {
      if (!(fc[i].bb_dist < max_len)) continue;                            /* ray.py:410 */
      int p = fc[i].prim, g = sc->prim_group[p];
      const double* M = sc->prim_xform + 12 * (size_t)p;
      int flags = sc->prim_flags[p];
      if (sc->prim_type[p] == ODW_PRIM_TRIANGLE) {
        /* one facet: same arithmetic as the default mode, finite-line test with line_len */
        v3 v0 = V(M[0], M[1], M[2]);
        v3 e1 = sub(V(M[3], M[4], M[5]), v0), e2 = sub(V(M[6], M[7], M[8]), v0);
        v3 nn = cross(e1, e2);
        double a2 = len(nn);
        v3 pv = cross(dn, e2);
        double det = dot(e1, pv);
        if (det == 0) continue;
        v3 tv = sub(start, v0);
        double u = dot(tv, pv) / det;
        v3 qv = cross(tv, e1);
        double v = dot(dn, qv) / det;
        int fe = sc->tri_edges ? sc->tri_edges[p] : 7;
        double a0 = (fe & 1) ? tol * (len(e2) / a2) : 1e-9;
        double a1 = (fe & 2) ? tol * (len(e1) / a2) : 1e-9;
        double a2e = (fe & 4) ? tol * (len(sub(e2, e1)) / a2) : 1e-9;
        if (u < -a0 || v < -a1 || u + v > 1.0 + a2e) continue;
        if ((u < 0 && (fe & 1)) || (v < 0 && (fe & 2)) || (u + v > 1.0 && (fe & 4))) {
          v3 w = add(mul(e1, u), mul(e2, v));
          double best = seg_dist2(w, e1);
          double other = seg_dist2(w, e2);
          if (other < best) best = other;
          other = seg_dist2(sub(w, e1), sub(e2, e1));
          if (other < best) best = other;
          if (best > tol * tol) continue;
        }
        double t = dot(e2, qv) / det;
        if (!(t > tol) || !(t < line_len + tol)) continue;
        v3 gp = add(start, mul(dn, t));
        v3 ng = mul(nn, 1.0 / a2);
        if (sc->tri_normals) {
          const double* vn = sc->tri_normals + 9 * (size_t)p;
          v3 tg = sub(gp, v0);
          double inv = 1.0 / dot(nn, nn);
          double bu = dot(cross(tg, e2), nn) * inv, bv = dot(cross(e1, tg), nn) * inv, bw = 1.0 - bu - bv;
          v3 ni = V(bw * vn[0] + bu * vn[3] + bv * vn[6], bw * vn[1] + bu * vn[4] + bv * vn[7],
                    bw * vn[2] + bu * vn[5] + bv * vn[8]);
          if (dot(ni, ni) > 0) ng = mul(ni, 1.0 / len(ni));
        }
        if (flags & ODW_FLAG_FLIP_NORMAL) ng = mul(ng, -1.0);
        if (n_hit == cap_hit) { cap_hit *= 2; hits = (nearest_hit*)realloc(hits, (size_t)cap_hit * sizeof(nearest_hit)); }
        nearest_hit* h = &hits[n_hit++];
        h->found = 1; h->prim = p; h->face = 0; h->group = g; h->dist = t; h->point = gp; h->normal = ng;
        max_len = t + 5 * tol;                                             /* ray.py:432 */
        continue;
      }
      v3 lstart = xf_point(M, start);
      v3 ldir = sub(xf_point(M, add(start, dn)), lstart);                  /* ray.py:348-349 */
      cand cs[8];
      int nc = prim_candidates(sc->prim_type[p], sc->prim_params + 4 * (size_t)p, 1 << fc[i].face,
                               lstart, ldir, tol, cs);
      for (int c = 0; c < nc; ++c) {
        double t = cs[c].t;
        v3 lp = add(lstart, mul(ldir, t));
        double dist = len(sub(lp, lstart));
        /* (vec-lstart).Length > distTol and vert.distToShape(line) < distTol: the point lies on the
         * infinite line, so its distance from the finite one is how far it is beyond either end */
        if (!(dist > tol)) continue;
        double beyond = t < 0 ? dist : (dist > line_len ? dist - line_len : 0.0);
        if (!(beyond < tol)) continue;
        v3 gp = xf_point_inv(M, lp);
        int ok = 1;
        for (int q = sc->prim_cond_off[p]; q < sc->prim_cond_off[p + 1] && ok; ++q) {
          int qp = sc->cond_prim[q];
          v3 pt = xf_point(sc->prim_xform + 12 * (size_t)qp, gp);
          double sd = prim_sdist(sc->prim_type[qp], sc->prim_params + 4 * (size_t)qp, pt);
          if (sc->cond_inside[q]) { if (sd > tol) ok = 0; }
          else { if (sd < -tol) ok = 0; }
        }
        if (!ok) continue;
        v3 nl = cs[c].n_local;
        if (flags & ODW_FLAG_FLIP_NORMAL) nl = mul(nl, -1.0);
        v3 ng = sub(xf_point_inv(M, add(lp, nl)), gp);
        if (n_hit == cap_hit) { cap_hit *= 2; hits = (nearest_hit*)realloc(hits, (size_t)cap_hit * sizeof(nearest_hit)); }
        nearest_hit* h = &hits[n_hit++];
        h->found = 1; h->prim = p; h->face = cs[c].face; h->group = g; h->dist = dist; h->point = gp; h->normal = ng;
        max_len = dist + 5 * tol;                                          /* ray.py:432 */
      }
    }
  }
  free(sh); free(fc);
  if (!n_hit) { free(hits); return none; }
  /* ray.py:438-452 */
  double min_dist = INFINITY;
  for (int i = 0; i < n_hit; ++i) if (hits[i].dist < min_dist) min_dist = hits[i].dist;
  int n_near = 0;
  for (int i = 0; i < n_hit; ++i) if (hits[i].dist < min_dist + 2 * tol) hits[n_near++] = hits[i];
  for (int i = 1; i < n_near; ++i) {            /* sorted(): stable */
    nearest_hit c = hits[i];
    int j = i - 1;
    while (j >= 0 && hits[j].dist > c.dist) { hits[j + 1] = hits[j]; --j; }
    hits[j + 1] = c;
  }
  nearest_hit res = hits[0];
  for (int i = 0; i < n_near; ++i) if (hits[i].group != medium) { res = hits[i]; break; }
  free(hits);
  return res;
}

/* ------------------------------------------------------------------ */
/* mirror / snellsLaw / lineGrating (ray.py:482-539)                   */
/* ------------------------------------------------------------------ */
static v3 mirror(v3 ray, v3 n) {
  /* -(2*normal*(ray*normal) - ray) */
  return mul(sub(mul(n, 2 * dot(ray, n)), ray), -1.0);
}

static v3 snells_law(v3 ray, double n1, double n2, v3 n, int* total_reflection) {
  v3 c = cross(n, ray);
  double s = n1 / n2 * n1 / n2;
  double root = 1 - dot(mul(c, s), c);
  if (root < 0) { *total_reflection = 1; return mirror(ray, n); }
  *total_reflection = 0;
  v3 inner = cross(mul(n, -1.0), ray);
  return add(mul(cross(n, inner), n1 / n2), mul(n, sqrt(root)));
}

static v3 line_grating(v3 ray, double n1, double n2, v3 normal, double wavelength_nm,
                       int order, double lpm, v3 gdir, int transmission) {
  double wavelength = wavelength_nm / 1000;
  ray = mul(ray, 1.0 / len(ray));
  v3 sn = mul(normal, 1.0 / len(normal));
  v3 g = mul(gdir, 1.0 / len(gdir));
  v3 P = cross(g, sn); P = mul(P, 1.0 / len(P));
  v3 D = cross(sn, P); D = mul(D, 1.0 / len(D));
  double mu = n1 / n2;
  double d = 1000 / lpm;
  double T = (order * wavelength) / (n1 * d);
  double V_ = (mu * dot(ray, sn)) / dot(sn, sn);
  double W = (mu * mu - 1 + T * T - 2 * mu * T * dot(ray, D)) / dot(sn, sn);
  double sq = sqrt((2 * V_) * (2 * V_) - 4 * W);
  double Q0 = (-2 * V_ + sq) / 2, Q1 = (-2 * V_ - sq) / 2;
  double Q = transmission ? fmin(Q0, Q1) : fmax(Q0, Q1);
  v3 S = add(sub(mul(ray, mu), mul(D, T)), mul(sn, Q));
  return mul(S, -1.0);
}

/* ------------------------------------------------------------------ */
/* OpticalGroupProxy.applyStochasticRayCorrections                     */
/* (optical_group.py:279-323) with the per-hit VectorRandomVariable    */
/* compile replaced by a family of pre-tabulated numeric-mode tables   */
/* (include/odw_trace.h, odw_surface_sampler_desc).                    */
/* ------------------------------------------------------------------ */
static const odw_surface_sampler_desc* g_samplers = NULL;
static int g_n_samplers = 0;
static uint64_t g_surface_seed = 0;

/* test-side registration: the descriptors (and their tables) must stay
 * alive until the next call; n = 0 clears */
int odw_oracle_set_surface_samplers(const odw_surface_sampler_desc* s, int32_t n, uint64_t explicit_ray_seed) {
  g_samplers = n > 0 ? s : NULL;
  g_n_samplers = n > 0 ? n : 0;
  g_surface_seed = explicit_ray_seed;
  return ODW_OK;
}

/* the sampler of (group, kind) that serves a hit with n1 / n2 = mu (mu <= 0: total reflection): mu = 0
 * samplers serve every hit, else the one nearest in log mu (include/odw_trace.h)                          */
static const odw_surface_sampler_desc* find_sampler(int group, int kind, double mu) {
  const odw_surface_sampler_desc* best = NULL;
  double dist = INFINITY;
  for (int i = 0; i < g_n_samplers; ++i) {
    const odw_surface_sampler_desc* S = &g_samplers[i];
    if (S->group != group || S->kind != kind) continue;
    if (S->mu == 0.0) return S;
    double d = (S->mu < 0 || mu <= 0) ? (((S->mu < 0) == (mu <= 0)) ? 0.0 : INFINITY) : fabs(log(S->mu / mu));
    if (!best || d < dist) { dist = d; best = S; }
  }
  return best;
}

/* FreeCAD Rotation(axis, angle) * v (Base::Rotation::setValue normalises the
 * axis, a zero axis leaves the vector unchanged) */
static v3 rotate(v3 axis, double angle, v3 v) {
  double l2 = dot(axis, axis);
  if (l2 == 0) return v;
  v3 k = mul(axis, 1.0 / sqrt(l2));
  double s = sin(angle), c = cos(angle);
  return add(add(mul(v, c), mul(cross(k, v), s)), mul(k, dot(k, v) * (1.0 - c)));
}

static double acos_clamped(double x) { return acos(fmax(-1.0, fmin(1.0, x))); }

static void surface_draw(const odw_surface_sampler_desc* S, double theta_in, double theta_refl,
                         uint64_t ray, uint64_t seed, uint32_t ordinal, uint32_t stream,
                         double* theta, double* phi) {
  int k = 0;
  if (S->family_axis != ODW_SURF_AXIS_NONE && S->n_family > 1) {
    double c = S->family_axis == ODW_SURF_AXIS_THETA_IN ? theta_in : theta_refl;
    double inv_step = (double)(S->n_family - 1) / (S->family_hi - S->family_lo);
    /* the two members around the hit's constant are mixed: member k0 + 1 with probability = the
     * fractional position between the knots, decided by a uniform of its own (stream + 16) */
    double kf = (c - S->family_lo) * inv_step;
    if (kf < 0) kf = 0;
    if (kf > (double)(S->n_family - 1)) kf = (double)(S->n_family - 1);
    int k0 = (int)floor(kf);
    double frac = kf - (double)k0;
    uint32_t mctr[4] = {(uint32_t)ray, (uint32_t)(ray >> 32), ordinal, stream + 16u};
    uint32_t mkey[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t mw[4];
    odw_oracle_philox(mctr, mkey, mw);
    k = k0 + (u53(mw[0], mw[1]) < frac ? 1 : 0);
  }
  uint32_t ctr[4] = {(uint32_t)ray, (uint32_t)(ray >> 32), ordinal, stream};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4];
  odw_oracle_philox(ctr, key, w);
  double u_phi = u53(w[0], w[1]), u_t = u53(w[2], w[3]);
  if (S->n_atoms) {
    /* discrete events: atom j owns [sum of the masses before it, + its own) of u_phi */
    double acc = 0.0;
    for (int j = 0; j < S->n_atoms; ++j) {
      double pj = S->atom_mass[(size_t)k * (size_t)S->n_atoms + j];
      if (u_phi < acc + pj) {
        const double* a = S->atom_theta + 3 * j;
        const double* b = S->atom_phi + 3 * j;
        *theta = a[0] + a[1] * theta_in + a[2] * theta_refl;
        *phi = b[0] != 0.0 ? b[1] + (u_phi - acc) / pj * (b[2] - b[1]) : b[1];
        return;
      }
      acc += pj;
    }
    u_phi = fmin((u_phi - acc) / (1.0 - acc), 0.99999999999999989);
  }
  odw_source_desc t;
  memset(&t, 0, sizeof t);
  t.n_phi_knots = S->n_phi_knots;
  t.phi_edges = S->phi_edges;
  t.phi_cdf = S->phi_cdf + (size_t)k * (size_t)S->n_phi_knots;
  t.n_t_knots = S->n_t_knots;
  t.n_t_rows = S->n_t_rows;
  t.t_edges = S->t_edges;
  t.t_cdf = S->t_cdf + (size_t)k * (size_t)S->n_t_rows * (size_t)S->n_t_knots;
  sample_one(&t, u_phi, u_t, theta, phi);
}

static v3 scatter(int group, uint64_t ray, uint64_t seed, uint32_t ordinal, v3 din, v3 ideal, v3 n, double mu) {
  const odw_surface_sampler_desc* prim = find_sampler(group, ODW_SURF_PRIMARY, mu);
  const odw_surface_sampler_desc* modi = find_sampler(group, ODW_SURF_MODIFY, mu);
  if (!prim && !modi) return ideal;
  double nl = len(n);
  double theta_in = acos_clamped(dot(din, n) / nl);
  double theta_refl = acos_clamped(dot(ideal, n) / (len(ideal) * nl));
  v3 out = ideal;
  double theta, phi;
  if (prim) {
    surface_draw(prim, theta_in, theta_refl, ray, seed, ordinal, 1u + ODW_SURF_PRIMARY, &theta, &phi);
    out = rotate(n, phi, rotate(cross(n, din), theta, n));
  }
  if (modi) {
    surface_draw(modi, theta_in, theta_refl, ray, seed, ordinal, 1u + ODW_SURF_MODIFY, &theta, &phi);
    out = rotate(out, phi, rotate(cross(out, din), theta, out));
  }
  return mul(out, 1.0 / len(out));
}

/* one applyStochasticRayCorrections call with explicit vectors (unit tests) */
int odw_oracle_scatter(int group, uint64_t ray, uint64_t seed, uint32_t ordinal, const double* din,
                       const double* ideal, const double* normal, double mu, double* out) {
  v3 r = scatter(group, ray, seed, ordinal, V(din[0], din[1], din[2]), V(ideal[0], ideal[1], ideal[2]),
                 V(normal[0], normal[1], normal[2]), mu);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
  return ODW_OK;
}

/* ------------------------------------------------------------------ */
/* Ray.traceRay (ray.py:36-281)                                        */
/* ------------------------------------------------------------------ */
typedef struct {
  odw_hit* hits; uint64_t cap; uint64_t n;
  uint64_t* hist;
  uint64_t cnt[ODW_CNT_COUNT];
  odw_segment* segs; uint64_t seg_cap; uint64_t seg_n;   /* grows */
} sink;

/* SimulationResultsSingleRay.addSegment (results_store.py:238-239) fed by the
 * tuples Ray.traceRay yields (ray.py:104-117) */
static void record_segment(sink* sk, uint64_t ray, int ordinal, int medium, v3 p1, v3 p2, double power) {
  if (sk->seg_n == sk->seg_cap) {
    sk->seg_cap = sk->seg_cap ? 2 * sk->seg_cap : 64;
    sk->segs = (odw_segment*)realloc(sk->segs, sk->seg_cap * sizeof(odw_segment));
  }
  odw_segment* g = &sk->segs[sk->seg_n++];
  g->p1[0] = p1.x; g->p1[1] = p1.y; g->p1[2] = p1.z;
  g->p2[0] = p2.x; g->p2[1] = p2.y; g->p2[2] = p2.z;
  g->power = power;
  g->tag = (ray & 0xFFFFFFFFFFull) | ((uint64_t)(ordinal & 0xFFF) << 40) | ((uint64_t)((medium + 1) & 0xFFF) << 52);
}

static void record_hit(sink* sk, const odw_detector_desc* det, uint32_t flags,
                       uint64_t ray, int group, v3 p, v3 d, double power, int entering) {
  sk->cnt[ODW_CNT_RECORDED_HITS]++;
  if (flags & ODW_TRACE_RECORD_HITS) {
    if (sk->n < sk->cap) {
      odw_hit* h = &sk->hits[sk->n];
      h->point[0] = p.x; h->point[1] = p.y; h->point[2] = p.z;
      h->direction[0] = d.x; h->direction[1] = d.y; h->direction[2] = d.z;
      h->power = power;
      h->tag = (ray & 0xFFFFFFFFFFFFull) | ((uint64_t)group << 48) | ((uint64_t)(entering != 0) << 63);
      sk->n++;
    } else {
      sk->cnt[ODW_CNT_HITS_DROPPED]++;
    }
  }
  if ((flags & ODW_TRACE_HISTOGRAM) && det && sk->hist && (det->group < 0 || det->group == group)) {
    v3 r = sub(p, V(det->origin[0], det->origin[1], det->origin[2]));
    double x = dot(r, V(det->ex[0], det->ex[1], det->ex[2]));
    double y = dot(r, V(det->ey[0], det->ey[1], det->ey[2]));
    double fx = floor((x - det->x_lo) * (det->nx / (det->x_hi - det->x_lo)));
    double fy = floor((y - det->y_lo) * (det->ny / (det->y_hi - det->y_lo)));
    if (fx >= 0 && fx < det->nx && fy >= 0 && fy < det->ny)
      sk->hist[(size_t)fx * (size_t)det->ny + (size_t)fy]++;
    else
      sk->cnt[ODW_CNT_HIST_OVERFLOW]++;
  }
}

static void trace_one(const odw_scene_desc* sc, const odw_limits* lim,
                      const odw_detector_desc* det, double wavelength, uint32_t flags,
                      uint64_t ray, uint64_t seed, v3 point, v3 dir, double power, sink* sk) {
  int seq = 0, nint = 0, medium = -1, skip_solid = -1;
  for (;;) {
    if (nint >= lim->max_intersections) { sk->cnt[ODW_CNT_CAPPED]++; break; }
    nint++;
    sk->cnt[ODW_CNT_SEGMENTS]++;
    nearest_hit h = g_strict ? nearest_strict(sc, lim, point, dir, medium, seq)
                             : nearest_skipping(sc, lim, point, dir, medium, seq, skip_solid);
    if (flags & ODW_TRACE_RECORD_SEGMENTS)
      record_segment(sk, ray, nint - 1, medium, point,
                     h.found ? h.point : add(point, mul(dir, lim->max_ray_length / len(dir))), power);
    if (!h.found) { sk->cnt[ODW_CNT_ESCAPED]++; break; }
    v3 prev = point;
    int prev_medium = medium;
    point = h.point;
    /* absorption in the medium just traversed (ray.py:120-125; assignment,
     * not multiplication, is the reference's behaviour) */
    if (prev_medium >= 0) {
      double L = sc->group_abslen[prev_medium];
      if (L == 0) power = 0;
      else if (isfinite(L)) power = exp(-len(sub(prev, point)) / L);
    }
    /* getNormal (ray.py:455-480) */
    v3 n = h.normal;
    v3 dray = sub(point, prev);
    double cosangle = dot(dray, n) / (len(dray) * len(n));
    int entering = cosangle < 0;
    if (entering) n = mul(n, -1.0);
    int g = h.group;
    if (sc->group_record[g]) record_hit(sk, det, flags, ray, g, point, dir, power, entering);

    int type = sc->group_type[g];
    if (type == ODW_OPT_MIRROR) {
      v3 din = mul(dir, 1.0 / len(dir));
      dir = scatter(g, ray, seed, (uint32_t)nint, din, mirror(dir, n), n, 1.0);
      power *= sc->group_refl[g];
      seq++;
    } else if (type == ODW_OPT_LENS) {
      double n1, n2;
      if (entering) {
        n1 = (medium >= 0) ? sc->group_ior[medium] : 1.0;
        medium = g;
        n2 = sc->group_ior[g];
      } else {
        n1 = (medium >= 0) ? sc->group_ior[medium] : 1.0;
        n2 = 1.0;
      }
      int tir;
      v3 din = mul(dir, 1.0 / len(dir));
      v3 ideal = snells_law(din, n1, n2, n, &tir);
      dir = scatter(g, ray, seed, (uint32_t)nint, din, ideal, n, tir ? -1.0 : n1 / n2);
      if (!entering && !tir && medium == g) { medium = -1; seq++; }
    } else if (type == ODW_OPT_GRATING) {
      v3 gd = V(sc->group_grating_dir[3 * g], sc->group_grating_dir[3 * g + 1],
                sc->group_grating_dir[3 * g + 2]);
      if (sc->group_grating_type[g] == 0) { /* reflection */
        if (entering) {
          double nn = (medium >= 0) ? sc->group_ior[medium] : 1.0;
          dir = line_grating(mul(dir, 1.0 / len(dir)), nn, nn, n, wavelength,
                             sc->group_grating_order[g], sc->group_grating_lpm[g], gd, 0);
          seq++;
        }
      } else { /* transmission */
        if (entering) {
          /* the reference raises ValueError if medium is not None; the host
           * validates scenes for that, here the ray is simply stopped */
          if (medium >= 0) { sk->cnt[ODW_CNT_GRATING_IN_MEDIUM]++; sk->cnt[ODW_CNT_DIED]++; break; }   /* ValueError, ray.py:234-237 */
          medium = g;
          dir = line_grating(mul(dir, 1.0 / len(dir)), 1.0, sc->group_ior[g], n, wavelength,
                             sc->group_grating_order[g], sc->group_grating_lpm[g], gd, 1);
        } else {
          double n1 = (medium >= 0) ? sc->group_ior[medium] : 1.0;
          int tir;
          dir = snells_law(mul(dir, 1.0 / len(dir)), n1, 1.0, n, &tir);
          if (!tir) { medium = -1; seq++; }
        }
      }
    } else if (type == ODW_OPT_ABSORBER) {
      power = 0;
      seq++;
    } else if (type == ODW_OPT_VACUUM) {
      seq++;
    }
    /* leaving a convex solid (outgoing direction along its outward normal): its
     * primitives are not candidates of the next segment */
    skip_solid = -1;
    if ((sc->prim_flags[h.prim] & ODW_FLAG_CONVEX) && sc->prim_solid) {
      if (sc->prim_type[h.prim] == ODW_PRIM_TRIANGLE) {
        /* a convex tessellated solid (scene/geometry.py: mesh_is_convex): the FACET's outward normal decides -- the
         * interpolated one of smooth shading can point out of the solid where the ray still runs into it */
        const double* M = sc->prim_xform + 12 * (size_t)h.prim;
        v3 v0 = V(M[0], M[1], M[2]);
        v3 ng = cross(sub(V(M[3], M[4], M[5]), v0), sub(V(M[6], M[7], M[8]), v0));
        double out = dot(dir, ng);
        if (sc->prim_flags[h.prim] & ODW_FLAG_FLIP_NORMAL) out = -out;
        if (out > 0) skip_solid = sc->prim_solid[h.prim];
      } else {
        double along = dot(dir, n);            /* n points along the incoming travel direction */
        if ((entering ? -along : along) > 0) skip_solid = sc->prim_solid[h.prim];
      }
    }
    if (power < lim->power_tol) { sk->cnt[ODW_CNT_DIED]++; break; }
  }
  sk->cnt[ODW_CNT_TRACED_RAYS]++;
}

/* ------------------------------------------------------------------ */
/* public entry points                                                 */
/* ------------------------------------------------------------------ */
#define CHUNK 4096

typedef struct { odw_segment* segs; uint64_t cap; uint64_t* n; } seg_out;

static int run(const odw_scene_desc* sc, const odw_source_desc* src, const odw_limits* lim,
               const odw_detector_desc* det, uint64_t first, uint64_t n, uint64_t seed,
               const double* origins, const double* dirs, const double* powers,
               uint32_t flags, odw_hit* hits, uint64_t cap, uint64_t* n_hits,
               uint64_t* hist, uint64_t* counters, int nthreads, const seg_out* so) {
  if (!sc || !lim || sc->n_groups > ODW_MAX_GROUPS) return ODW_ERR_INVALID;
  if (!so) flags &= ~(uint32_t)ODW_TRACE_RECORD_SEGMENTS;
  odw_segment** chunk_segs = NULL;
  uint64_t* chunk_seg_n = NULL;
  if (!origins && !src) return ODW_ERR_INVALID;
  uint64_t nchunks = (n + CHUNK - 1) / CHUNK;
  size_t nbins = det ? (size_t)det->nx * (size_t)det->ny : 0;
  /* per-chunk private hit lists keep the output in ray order under OpenMP */
  odw_hit** chunk_hits = (odw_hit**)calloc(nchunks ? nchunks : 1, sizeof(odw_hit*));
  uint64_t* chunk_n = (uint64_t*)calloc(nchunks ? nchunks : 1, sizeof(uint64_t));
  if (so) {
    chunk_segs = (odw_segment**)calloc(nchunks ? nchunks : 1, sizeof(odw_segment*));
    chunk_seg_n = (uint64_t*)calloc(nchunks ? nchunks : 1, sizeof(uint64_t));
  }
  uint64_t total[ODW_CNT_COUNT] = {0};
  int max_hits_per_ray = lim->max_intersections > 0 ? lim->max_intersections : 1;
  (void)nthreads;
  if (g_strict) strict_prepare(sc, lim->dist_tol);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    uint64_t* lhist = (nbins && hist && (flags & ODW_TRACE_HISTOGRAM)) ? (uint64_t*)calloc(nbins, sizeof(uint64_t)) : NULL;
    uint64_t lcnt[ODW_CNT_COUNT] = {0};
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t ci = 0; ci < (int64_t)nchunks; ++ci) {
      uint64_t b = (uint64_t)ci * CHUNK, e = b + CHUNK < n ? b + CHUNK : n;
      sink sk;
      memset(&sk, 0, sizeof sk);
      sk.hist = lhist;
      if (flags & ODW_TRACE_RECORD_HITS) {
        sk.cap = 64; /* grows */
        sk.hits = (odw_hit*)malloc(sk.cap * sizeof(odw_hit));
      }
      for (uint64_t i = b; i < e; ++i) {
        if ((flags & ODW_TRACE_RECORD_HITS) && sk.cap - sk.n < (uint64_t)max_hits_per_ray) {
          sk.cap = sk.cap * 2 + (uint64_t)max_hits_per_ray;
          sk.hits = (odw_hit*)realloc(sk.hits, sk.cap * sizeof(odw_hit));
        }
        v3 o, d;
        double pw;
        if (origins) {
          o = V(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
          d = V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
          pw = powers ? powers[i] : 1.0;
        } else {
          double up, ut, t, phi;
          ray_uniforms(first + i, seed, &up, &ut);
          sample_one(src, up, ut, &t, &phi);
          make_ray(src, t, phi, &o, &d);
          pw = src->power;
        }
        trace_one(sc, lim, det, src ? src->wavelength : 500.0, flags, first + i,
                  origins ? g_surface_seed : seed, o, d, pw, &sk);
      }
      chunk_hits[ci] = sk.hits;
      chunk_n[ci] = sk.n;
      if (so) { chunk_segs[ci] = sk.segs; chunk_seg_n[ci] = sk.seg_n; }
      for (int k = 0; k < ODW_CNT_COUNT; ++k) lcnt[k] += sk.cnt[k];
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      for (int k = 0; k < ODW_CNT_COUNT; ++k) total[k] += lcnt[k];
      if (lhist) for (size_t i = 0; i < nbins; ++i) hist[i] += lhist[i];
    }
    free(lhist);
  }
  uint64_t nh = 0, dropped = 0;
  for (uint64_t ci = 0; ci < nchunks; ++ci) {
    for (uint64_t k = 0; k < chunk_n[ci]; ++k) {
      if (hits && nh < cap) hits[nh++] = chunk_hits[ci][k];
      else dropped++;
    }
    free(chunk_hits[ci]);
  }
  free(chunk_hits); free(chunk_n);
  if (so) {   /* per-chunk lists are in (ray, ordinal) order already */
    uint64_t ns = 0;
    for (uint64_t ci = 0; ci < nchunks; ++ci) {
      for (uint64_t k = 0; k < chunk_seg_n[ci]; ++k)
        if (so->segs && ns < so->cap) so->segs[ns++] = chunk_segs[ci][k];
      free(chunk_segs[ci]);
    }
    free(chunk_segs); free(chunk_seg_n);
    if (so->n) *so->n = ns;
  }
  total[ODW_CNT_HITS_DROPPED] += dropped;
  if (n_hits) *n_hits = nh;
  if (counters) for (int k = 0; k < ODW_CNT_COUNT; ++k) counters[k] += total[k];
  return ODW_OK;
}

int odw_oracle_trace(const odw_scene_desc* sc, const odw_source_desc* src,
                     const odw_limits* lim, const odw_detector_desc* det,
                     uint64_t first, uint64_t n, uint64_t seed, uint32_t flags,
                     odw_hit* hits, uint64_t cap, uint64_t* n_hits, uint64_t* hist,
                     uint64_t* counters, int nthreads) {
  return run(sc, src, lim, det, first, n, seed, NULL, NULL, NULL, flags, hits, cap, n_hits,
             hist, counters, nthreads, NULL);
}

int odw_oracle_trace_rays(const odw_scene_desc* sc, const odw_limits* lim,
                          const odw_detector_desc* det, double wavelength,
                          uint64_t first, uint64_t n, const double* origins,
                          const double* dirs, const double* powers, uint32_t flags,
                          odw_hit* hits, uint64_t cap, uint64_t* n_hits, uint64_t* hist,
                          uint64_t* counters, int nthreads) {
  odw_source_desc s;
  memset(&s, 0, sizeof s);
  s.wavelength = wavelength;
  s.power = 1.0;
  return run(sc, &s, lim, det, first, n, 0, origins, dirs, powers, flags, hits, cap, n_hits,
             hist, counters, nthreads, NULL);
}

/* the segments of every ray (RecordRays, generic_source.py:78-118), sorted by
 * (ray, ordinal); origins == NULL: rays of the sampler `src`, else explicit
 * initial conditions with the wavelength given */
int odw_oracle_trace_segments(const odw_scene_desc* sc, const odw_source_desc* src,
                              const odw_limits* lim, double wavelength, uint64_t first, uint64_t n,
                              uint64_t seed, const double* origins, const double* dirs,
                              const double* powers, odw_segment* segs, uint64_t cap,
                              uint64_t* n_segs, uint64_t* counters) {
  seg_out so = {segs, cap, n_segs};
  if (origins) {
    odw_source_desc s;
    memset(&s, 0, sizeof s);
    s.wavelength = wavelength;
    s.power = 1.0;
    return run(sc, &s, lim, NULL, first, n, 0, origins, dirs, powers, ODW_TRACE_RECORD_SEGMENTS, NULL, 0,
               NULL, NULL, counters, 0, &so);
  }
  return run(sc, src, lim, NULL, first, n, seed, NULL, NULL, NULL, ODW_TRACE_RECORD_SEGMENTS, NULL, 0,
             NULL, NULL, counters, 0, &so);
}

/* initial conditions only: origin/direction of sampled rays (pins _makeRay) */
int odw_oracle_make_rays(const odw_source_desc* src, uint64_t first, uint64_t n,
                         uint64_t seed, double* origins, double* dirs) {
  for (uint64_t i = 0; i < n; ++i) {
    double up, ut, t, phi;
    v3 o, d;
    ray_uniforms(first + i, seed, &up, &ut);
    sample_one(src, up, ut, &t, &phi);
    make_ray(src, t, phi, &o, &d);
    origins[3 * i] = o.x; origins[3 * i + 1] = o.y; origins[3 * i + 2] = o.z;
    dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
  }
  return ODW_OK;
}

/* ------------------------------------------------------------------ */
/* SurfaceSourceProxy._generateRays(mode='true')                       */
/* (surface_source.py:519-553, _makeRay :87-108) on analytic faces;    */
/* sampling rule and Philox counters: include/odw_trace.h,             */
/* odw_surface_source_desc.                                            */
/* ------------------------------------------------------------------ */
static void philox_pair(uint64_t ray, uint64_t seed, uint32_t c2, uint32_t c3, double* a, double* b) {
  uint32_t ctr[4] = {(uint32_t)ray, (uint32_t)(ray >> 32), c2, c3};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t w[4];
  odw_oracle_philox(ctr, key, w);
  *a = u53(w[0], w[1]);
  *b = u53(w[2], w[3]);
}

static double face_point(int type, const double* par, int face, double ua, double ub, v3* p, v3* n, v3* t) {
  const double two_pi = 6.283185307179586;
  if (type == ODW_PRIM_BOX) {
    int a = face >> 1, b1 = (a + 1) % 3, b2 = (a + 2) % 3;
    double c[3], nn[3] = {0, 0, 0}, tt[3] = {0, 0, 0};
    c[a] = (face & 1) ? par[a] : 0.0;
    c[b1] = ua * par[b1];
    c[b2] = ub * par[b2];
    nn[a] = (face & 1) ? 1.0 : -1.0;
    tt[b1] = 1.0;
    *p = V(c[0], c[1], c[2]); *n = V(nn[0], nn[1], nn[2]); *t = V(tt[0], tt[1], tt[2]);
    return 1.0;
  }
  double sa = sin(two_pi * ua), ca = cos(two_pi * ua);
  if (type == ODW_PRIM_SPHERE) {
    double z = 2.0 * ub - 1.0, r = sqrt(fmax(0.0, 1.0 - z * z));
    *n = V(r * ca, r * sa, z);
    *p = mul(*n, par[0]);
    *t = V(-sa, ca, 0.0);
    return 1.0;
  }
  if (type == ODW_PRIM_TORUS) {
    double sv = sin(two_pi * ub), cv = cos(two_pi * ub);
    double rho = par[0] + par[1] * cv;
    *p = V(rho * ca, rho * sa, par[1] * sv);
    *n = V(cv * ca, cv * sa, sv);
    *t = V(-sa, ca, 0.0);
    return rho / (par[0] + par[1]);
  }
  double r1 = par[0], r2 = (type == ODW_PRIM_CONE) ? par[1] : par[0];
  double h = (type == ODW_PRIM_CONE) ? par[2] : par[1];
  if (face == 0) {
    double z;
    if (r1 == r2) z = h * ub;
    else z = h * (sqrt(r1 * r1 + ub * (r2 * r2 - r1 * r1)) - r1) / (r2 - r1);
    double k = (r2 - r1) / h, r = r1 + k * z;
    double inv = 1.0 / sqrt(1.0 + k * k);
    *p = V(r * ca, r * sa, z);
    *n = V(ca * inv, sa * inv, -k * inv);
    *t = V(-sa, ca, 0.0);
    return 1.0;
  }
  double rr = (face == 1 ? r1 : r2) * sqrt(ub);
  *p = V(rr * ca, rr * sa, face == 1 ? 0.0 : h);
  *n = V(0.0, 0.0, face == 1 ? -1.0 : 1.0);
  *t = V(1.0, 0.0, 0.0);
  return 1.0;
}

static v3 xf_vec_inv(const double* m, v3 v) { /* R^T v */
  return V(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
           m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

int odw_oracle_surface_rays(const odw_surface_source_desc* s, uint64_t first, uint64_t n, uint64_t seed,
                            double* origins, double* dirs) {
  if (!s || s->n_faces < 1) return ODW_ERR_INVALID;
  double total = 0;
  for (int f = 0; f < s->n_faces; ++f) total += s->face_area[f];
  double* cdf = (double*)malloc(((size_t)s->n_faces + 1) * sizeof(double));
  double run = 0;
  cdf[0] = 0;
  for (int f = 0; f < s->n_faces; ++f) { run += s->face_area[f]; cdf[f + 1] = run / total; }
  cdf[s->n_faces] = 1.0;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t ray = first + i;
    v3 gp = V(0, 0, 0), gn = V(0, 0, 1), gt = V(1, 0, 0);
    for (uint32_t attempt = 0; attempt < 4096; ++attempt) {
      double u_face, u_acc, ua, ub;
      philox_pair(ray, seed, attempt, 3u, &u_face, &u_acc);
      philox_pair(ray, seed, attempt, 4u, &ua, &ub);
      int f = 0, hi = s->n_faces - 1;         /* largest f with cdf[f] <= u_face */
      while (f < hi) {
        int mid = (f + hi + 1) >> 1;
        if (u_face >= cdf[mid]) f = mid; else hi = mid - 1;
      }
      int prim = s->face_prim[f], face = s->face_id[f];
      const double* m = s->prim_xform + 12 * (size_t)prim;
      if (s->prim_type[prim] == ODW_PRIM_TRIANGLE) {     /* a facet of a tessellated face */
        v3 v0 = V(m[0], m[1], m[2]), e1 = sub(V(m[3], m[4], m[5]), v0), e2 = sub(V(m[6], m[7], m[8]), v0);
        double a = ua, b = ub;
        if (a + b > 1.0) { a = 1.0 - a; b = 1.0 - b; }
        gp = add(v0, add(mul(e1, a), mul(e2, b)));
        v3 fn = cross(e1, e2);
        gn = mul(fn, 1.0 / len(fn));
        if (s->tri_normals) {
          const double* vn = s->tri_normals + 9 * (size_t)prim;
          v3 mix = add(mul(V(vn[0], vn[1], vn[2]), 1.0 - a - b),
                       add(mul(V(vn[3], vn[4], vn[5]), a), mul(V(vn[6], vn[7], vn[8]), b)));
          gn = mul(mix, 1.0 / len(mix));
        }
        if (s->prim_flags[prim] & ODW_FLAG_FLIP_NORMAL) gn = mul(gn, -1.0);
        gt = sub(e1, mul(gn, dot(e1, gn)));
        gt = mul(gt, 1.0 / len(gt));
        break;
      }
      v3 p, nl, tl;
      double accept = face_point(s->prim_type[prim], s->prim_params + 4 * (size_t)prim, face, ua, ub, &p, &nl, &tl);
      if (u_acc >= accept) continue;
      if (s->prim_flags[prim] & ODW_FLAG_FLIP_NORMAL) nl = mul(nl, -1.0);
      gp = xf_vec_inv(m, V(p.x - m[3], p.y - m[7], p.z - m[11]));
      int ok = 1;
      for (int c = s->prim_cond_off[prim]; c < s->prim_cond_off[prim + 1] && ok; ++c) {
        int qp = s->cond_prim[c];
        double sd = prim_sdist(s->prim_type[qp], s->prim_params + 4 * (size_t)qp,
                               xf_point(s->prim_xform + 12 * (size_t)qp, gp));
        if (s->cond_inside[c]) { if (sd > s->dist_tol) ok = 0; }
        else { if (sd < -s->dist_tol) ok = 0; }
      }
      if (!ok) continue;
      gn = xf_vec_inv(m, nl);
      gt = xf_vec_inv(m, tl);
      break;
    }
    double u_t, u_phi;
    philox_pair(ray, seed, 0u, 5u, &u_t, &u_phi);
    double theta = odw_oracle_interp(u_t, s->t_cdf, s->t_edges, s->n_t_knots);
    double phi = 6.283185307179586 * u_phi;
    v3 d = rotate(gn, phi, rotate(gt, theta, gn));
    d = mul(d, 1.0 / len(d));
    origins[3 * i] = gp.x; origins[3 * i + 1] = gp.y; origins[3 * i + 2] = gp.z;
    dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
  }
  free(cdf);
  return ODW_OK;
}

/* single nearest-hit query, for unit tests of the geometry */
int odw_oracle_nearest(const odw_scene_desc* sc, const odw_limits* lim, const double* start,
                       const double* dir, int medium, int seq_idx, int* prim, int* face,
                       int* group, double* dist, double* point, double* normal) {
  if (g_strict) strict_prepare(sc, lim->dist_tol);
  nearest_hit h = g_strict ? nearest_strict(sc, lim, V(start[0], start[1], start[2]), V(dir[0], dir[1], dir[2]), medium, seq_idx)
                           : nearest(sc, lim, V(start[0], start[1], start[2]), V(dir[0], dir[1], dir[2]),
                                     medium, seq_idx);
  if (!h.found) return 0;
  *prim = h.prim; *face = h.face; *group = h.group; *dist = h.dist;
  point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
  normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
  return 1;
}

int odw_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
