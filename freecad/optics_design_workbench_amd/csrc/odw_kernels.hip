// odw_kernels.hip -- hand-written gfx950 kernels of the Monte-Carlo hot path.
//
// One thread traces one ray from generation to termination ("megakernel"):
//   K1 generate   Philox4x32-10(ray index) -> inverse-CDF tables -> ray
//                 (reference: random_number_generator.py:467-560,
//                  point_source.py:411-460, 659-679)
//   K2 nearest    analytic line/surface intersection over the baked
//                 primitives with the reference's tolerance rules
//                 (ray.py:290-452)
//   K3 interact   normal, entering test, mirror / Snell / absorb / vacuum /
//                 grating, medium + sequence state (ray.py:91-281, 455-539)
//   K5 record     wave-aggregated append of 64-B hit rows (ballot + popcount
//                 prefix, one atomic per wave) and u64 histogram scatter
//                 (optical_group.py:206-209 -> results_store.py:641-648)
// Ray state lives in registers for the whole life of the ray; the scene is
// read through wave-uniform (scalar) loads.  float64 throughout, like the
// reference (FreeCAD Vector/Matrix are double).
#include "odw_device.h"

namespace odw {

// Scene tables are immutable during a launch.  Reading them through the
// constant address space lets hipcc use scalar loads (s_load_*) whenever the
// index is wave-uniform -- which it is in the flat primitive loop -- so the
// scene occupies SGPRs / the scalar cache instead of 16 VGPR pairs per
// primitive.  (Plain global pointers inside a by-value struct are not
// provably unclobbered, and hipcc falls back to per-lane global_load.)
#define ODW_CONST __attribute__((address_space(4)))
typedef const double ODW_CONST* cf64;
typedef const int32_t ODW_CONST* ci32;
typedef const uint64_t ODW_CONST* cu64;
template <class T>
__device__ __forceinline__ const T ODW_CONST* as_const(const T* p) {
  return (const T ODW_CONST*)(uintptr_t)p;
}

// ---- cheap float64 reciprocal / square root ------------------------------
// v_rcp_f64 / v_rsq_f64 deliver ~26 bits; two Newton steps give ~1 ulp without
// the v_div_scale/v_div_fmas/v_div_fixup tail of an IEEE division (used where
// the result feeds tolerance tests with 1e-6 slack, never in the sampler).
__device__ __forceinline__ double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return x > 0 ? g : 0.0;
}
__device__ __forceinline__ double frsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// ---------------------------------------------------------------- Philox
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2,
                                              uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void ray_uniforms(uint64_t ray, uint64_t seed, double& u_phi,
                                             double& u_t) {
  uint32_t c0 = (uint32_t)ray, c1 = (uint32_t)(ray >> 32), c2 = 0u, c3 = 0u;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
  u_phi = u53(c0, c1);
  u_t = u53(c2, c3);
}

// ------------------------------------------------ inverse CDF (numpy.interp)
// table = interleaved (cdf, edge) pairs; cdf[0] = 0, cdf[n-1] = 1, u in [0,1).
// Finds j = last knot with cdf[j] <= u inside the bracket [lo, hi] and
// evaluates slope*(u - cdf[j]) + edge[j] WITHOUT fma contraction, i.e. the
// exact arithmetic of numpy's arr_interp.
__device__ __forceinline__ double inv_cdf(const double* __restrict__ tab, int lo, int hi, double u) {
#pragma clang fp contract(off)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (u >= tab[2 * mid]) lo = mid; else hi = mid;
  }
  const double2 a = *reinterpret_cast<const double2*>(tab + 2 * lo);
  const double2 b = *reinterpret_cast<const double2*>(tab + 2 * lo + 2);
  if (a.x == u) return a.y;
  const double slope = (b.y - a.y) / (b.x - a.x);
  const double prod = slope * (u - a.x);
  return prod + a.y;
}

__device__ __forceinline__ void sample_source(const DeviceSource& s, double u_phi, double u_t,
                                              double& t_out, double& phi_out) {
#pragma clang fp contract(off)
  const double phi = inv_cdf(s.phi_tab, 0, s.n_phi_knots - 1, u_phi);
  int row = 0;
  if (s.n_t_rows > 1) {
    // argmin_i |mid_i - phi| (first minimum): the candidate is the cell that
    // contains phi, the exact rule is applied to it and its two neighbours
    const double e0 = s.phi_tab[1], e1 = s.phi_tab[2 * (s.n_phi_knots - 1) + 1];
    int r = (int)floor((phi - e0) / (e1 - e0) * (double)s.n_t_rows);
    r = max(0, min(s.n_t_rows - 1, r));
    double best = INFINITY;
    for (int i = max(0, r - 1); i <= min(s.n_t_rows - 1, r + 1); ++i) {
      const double mid = (s.phi_tab[2 * (i + 1) + 1] + s.phi_tab[2 * i + 1]) / 2.0;
      const double d = fabs(mid - phi);
      if (d < best) { best = d; row = i; }
    }
  }
  const double* tab = s.t_tab + (size_t)row * (size_t)s.n_t_knots * 2;
  const int32_t* guide = s.t_guide + (size_t)row * (size_t)(s.n_guide + 1);
  const int k = (int)(u_t * (double)s.n_guide);
  const int lo = guide[k];
  const int hi = min(guide[k + 1] + 1, s.n_t_knots - 1);
  t_out = inv_cdf(tab, lo, hi, u_t);
  phi_out = phi;
}

template <class P>
__device__ __forceinline__ d3 xf_point(P m, d3 p) {
  return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
            m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <class P>
__device__ __forceinline__ d3 xf_vec(P m, d3 v) {
  return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
template <class P>
__device__ __forceinline__ d3 xf_vec_t(P m, d3 v) {  // R^T v
  return mk(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
            m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

// PointSourceProxy._makeRay (point_source.py:411-460)
__device__ __forceinline__ void make_ray(const DeviceSource& s, double t_or_r, double phi,
                                         d3& origin, d3& dir) {
  d3 ldir, lorg;
  double sp, cp;
  sincos(phi, &sp, &cp);
  if (s.finite_focal) {
    double st, ct;
    sincos(t_or_r, &st, &ct);
    ldir = mk(st * sp, -st * cp, ct);
    lorg = (mk(0, 0, 1) - ldir) * s.focal_length;
  } else {
    ldir = mk(0, 0, 1);
    lorg = mk(t_or_r * cp, -t_or_r * sp, 0.0);
  }
  const d3 ln = ldir * (1.0 / sqrt(dot(ldir, ldir)));
  const d3 p1 = xf_point(s.m, lorg);
  const d3 p2 = xf_point(s.m, lorg + ln);
  const d3 d = p2 - p1;
  origin = p1;
  dir = d * (1.0 / sqrt(dot(d, d)));
}

// ----------------------------------------------------------- primitives
__device__ __forceinline__ double prim_sdist(int type, cf64 par, d3 p) {
  switch (type) {
    case ODW_PRIM_BOX: {
      const double dx = fmax(-p.x, p.x - par[0]);
      const double dy = fmax(-p.y, p.y - par[1]);
      const double dz = fmax(-p.z, p.z - par[2]);
      return fmax(dx, fmax(dy, dz));
    }
    case ODW_PRIM_SPHERE: return fsqrt(dot(p, p)) - par[0];
    case ODW_PRIM_CYLINDER: {
      const double rho = fsqrt(p.x * p.x + p.y * p.y);
      return fmax(rho - par[0], fmax(-p.z, p.z - par[1]));
    }
    case ODW_PRIM_CONE: {
      const double k = (par[1] - par[0]) / par[2];
      const double rho = fsqrt(p.x * p.x + p.y * p.y);
      const double lat = (rho - (par[0] + k * p.z)) * frsqrt(1 + k * k);
      return fmax(lat, fmax(-p.z, p.z - par[2]));
    }
    default: {  // torus
      const double rho = fsqrt(p.x * p.x + p.y * p.y);
      const double a = rho - par[0];
      return fsqrt(a * a + p.z * p.z) - par[1];
    }
  }
}

// a t^2 + 2 bh t + c = 0, cancellation-free; returns number of roots, t0 <= t1
__device__ __forceinline__ int quad_roots(double a, double bh, double c, double& t0, double& t1) {
  if (a == 0) {
    if (bh == 0) return 0;
    t0 = t1 = -c * frcp(2 * bh);
    return 1;
  }
  const double disc = bh * bh - a * c;
  if (!(disc >= 0)) return 0;
  const double sq = fsqrt(disc);
  const double q = -(bh + (bh >= 0 ? sq : -sq));
  double r0 = q * frcp(a);
  double r1 = (q != 0) ? c * frcp(q) : r0;
  if (r0 > r1) { const double tmp = r0; r0 = r1; r1 = tmp; }
  t0 = r0; t1 = r1;
  return 2;
}
// the same for a == 1 (unit direction in a rigid frame)
__device__ __forceinline__ int quad_roots_unit(double bh, double c, double& t0, double& t1) {
  const double disc = bh * bh - c;
  if (!(disc >= 0)) return 0;
  const double sq = fsqrt(disc);
  const double q = -(bh + (bh >= 0 ? sq : -sq));
  double r0 = q;
  double r1 = (q != 0) ? c * frcp(q) : r0;
  if (r0 > r1) { const double tmp = r0; r0 = r1; r1 = tmp; }
  t0 = r0; t1 = r1;
  return 2;
}

// ---- torus quartic: roots by derivative isolation + safeguarded Newton ----
// Rare (only when a ray really enters the torus' bounding slab outside the
// hole), kept out of line so its private arrays do not cost the hot loop
// registers or scratch traffic.
__device__ __forceinline__ double poly_eval(const double* c, int deg, double t) {
  double r = c[deg];
  for (int i = deg - 1; i >= 0; --i) r = r * t + c[i];
  return r;
}

__device__ __noinline__ double mono_root(const double* c, int deg, double a, double b) {
  double dc[4];
  for (int i = 1; i <= deg; ++i) dc[i - 1] = c[i] * i;
  const double fa = poly_eval(c, deg, a);
  double lo = a, hi = b;
  if (fa > 0) { lo = b; hi = a; }
  double x = 0.5 * (a + b);
  for (int it = 0; it < 200; ++it) {
    const double f = poly_eval(c, deg, x);
    if (f == 0) return x;
    if (f < 0) lo = x; else hi = x;
    const double df = poly_eval(dc, deg - 1, x);
    double xn = (df != 0) ? x - f / df : 0.5 * (lo + hi);
    const double mn = fmin(lo, hi), mx = fmax(lo, hi);
    if (!(xn > mn && xn < mx)) xn = 0.5 * (lo + hi);
    if (xn == x || fabs(hi - lo) <= 4e-16 * (fabs(lo) + fabs(hi))) return xn;
    x = xn;
  }
  return x;
}

// sign-change roots of c (degree deg) on the pieces between breakpoints
__device__ __noinline__ int roots_between(const double* c, int deg, const double* brk, int nb, double* out) {
  int n = 0;
  for (int i = 0; i + 1 < nb; ++i) {
    const double a = brk[i], b = brk[i + 1];
    if (!(b > a)) continue;
    const double fa = poly_eval(c, deg, a), fb = poly_eval(c, deg, b);
    if (fa == 0) {
      if (n == 0 || out[n - 1] != a) out[n++] = a;
      continue;
    }
    if ((fa < 0 && fb > 0) || (fa > 0 && fb < 0)) out[n++] = mono_root(c, deg, a, b);
    else if (fb == 0 && i + 2 == nb) out[n++] = b;
  }
  return n;
}

// smallest-first roots of the torus quartic on [lo, hi]; the coefficients are
// rebuilt here from the 8 scalars the caller has in registers
__device__ __noinline__ int torus_roots(double A, double B, double C, double E, double F, double G,
                                        double R1, double lo, double hi, double* out) {
  double c4[5];
  c4[4] = A * A;
  c4[3] = 2 * A * B;
  c4[2] = B * B + 2 * A * C - 4 * R1 * R1 * E;
  c4[1] = 2 * B * C - 4 * R1 * R1 * F;
  c4[0] = C * C - 4 * R1 * R1 * G;
  // q'' (quadratic) -> q' (cubic) -> q (quartic)
  double c3[4] = {c4[1], 2 * c4[2], 3 * c4[3], 4 * c4[4]};
  double c2[3] = {c3[1], 2 * c3[2], 3 * c3[3]};
  double brk[6];
  int nb = 0;
  brk[nb++] = lo;
  {
    double r0, r1;
    const int nr = quad_roots(c2[2], 0.5 * c2[1], c2[0], r0, r1);
    if (nr >= 1 && r0 > lo && r0 < hi) brk[nb++] = r0;
    if (nr == 2 && r1 > lo && r1 < hi && r1 != r0) brk[nb++] = r1;
  }
  brk[nb++] = hi;
  double crit[4];
  const int nc = roots_between(c3, 3, brk, nb, crit);
  nb = 0;
  brk[nb++] = lo;
  for (int i = 0; i < nc; ++i) brk[nb++] = crit[i];
  brk[nb++] = hi;
  return roots_between(c4, 4, brk, nb, out);
}

// ------------------------------------------------------------------------
struct Best {
  double t;
  int prim, face;
};
__device__ __forceinline__ bool better(double t, int p, int f, const Best& b) {
  if (t != b.t) return t < b.t;
  if (p != b.prim) return p < b.prim;
  return f < b.face;
}

struct Query {
  d3 start, dn;       // global ray (unit direction)
  double tol, tmax;   // distTol, maxRayLength + distTol
  int medium;
  Best any, oth;
};

struct SceneView {    // constant-address-space views of the scene tables
  cf64 prim_f64, prim_box;
  ci32 prim_i32, cond_i32;
};

// trimming by the other operands of a boolean (cond list), then bookkeeping
// of the two running minima (nearest of all / nearest not in current medium)
__device__ __forceinline__ void consider(const SceneView& sv, Query& q, double t, int p, int face,
                                         int group, int cond_off, int cond_cnt) {
  if (!(t > q.tol && t < q.tmax)) return;
  const bool cand_any = better(t, p, face, q.any);
  const bool cand_oth = (group != q.medium) && better(t, p, face, q.oth);
  if (!cand_any && !cand_oth) return;
  if (cond_cnt) {
    const d3 gp = q.start + q.dn * t;
    for (int c = cond_off; c < cond_off + cond_cnt; ++c) {
      const int cw = sv.cond_i32[c];
      const int qp = cw & 0x7fffffff;
      cf64 pf = sv.prim_f64 + (size_t)qp * 16;
      const double sd = prim_sdist(sv.prim_i32[4 * qp], pf + 12, xf_point(pf, gp));
      if (cw < 0) { if (sd > q.tol) return; }     // must be inside
      else { if (sd < -q.tol) return; }           // must be outside
    }
  }
  if (cand_any) { q.any.t = t; q.any.prim = p; q.any.face = face; }
  if (cand_oth) { q.oth.t = t; q.oth.prim = p; q.oth.face = face; }
}

// every face of primitive p against the ray: untrimmed analytic surface,
// natural face bounds with tolerance (ray.py:411-426)
__device__ __forceinline__ void intersect_prim(const SceneView& sv, Query& q, int p) {
  cf64 pf = sv.prim_f64 + (size_t)p * 16;
  ci32 pi = sv.prim_i32 + 4 * p;
  const int type = pi[0], group = pi[1], flags = pi[2];
  const int cond_off = pi[3] & 0xffffff, cond_cnt = (pi[3] >> 24) & 0xff;
  const int fmask = flags >> ODW_FACEMASK_SHIFT;
  cf64 par = pf + 12;
  const double tol = q.tol;
  const d3 o = xf_point(pf, q.start);
  const d3 d = xf_vec(pf, q.dn);

  if (type == ODW_PRIM_BOX) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!((fmask >> (2 * a)) & 3)) continue;
      const int b1 = (a + 1) % 3, b2 = (a + 2) % 3;
      const double oa = comp(o, a), da = comp(d, a);
      const double inv = frcp(da);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!((fmask >> (2 * a + s)) & 1)) continue;
        const double t = ((s ? par[a] : 0.0) - oa) * inv;
        const double p1 = comp(o, b1) + t * comp(d, b1);
        const double p2 = comp(o, b2) + t * comp(d, b2);
        if (p1 >= -tol && p1 <= par[b1] + tol && p2 >= -tol && p2 <= par[b2] + tol)
          consider(sv, q, t, p, 2 * a + s, group, cond_off, cond_cnt);
      }
    }
  } else if (type == ODW_PRIM_SPHERE) {
    if (fmask & 1) {
      double t0, t1;
      const int nr = quad_roots_unit(dot(o, d), dot(o, o) - par[0] * par[0], t0, t1);
      if (nr == 2) {
        consider(sv, q, t0, p, 0, group, cond_off, cond_cnt);
        consider(sv, q, t1, p, 0, group, cond_off, cond_cnt);
      }
    }
  } else if (type == ODW_PRIM_CYLINDER || type == ODW_PRIM_CONE) {
    const double R1 = par[0];
    const double R2 = (type == ODW_PRIM_CYLINDER) ? par[0] : par[1];
    const double H = (type == ODW_PRIM_CYLINDER) ? par[1] : par[2];
    const double k = (type == ODW_PRIM_CYLINDER) ? 0.0 : (R2 - R1) / H;
    if (fmask & 1) {
      const double rz = R1 + k * o.z;
      double t0, t1;
      const int nr = quad_roots(d.x * d.x + d.y * d.y - k * k * d.z * d.z,
                                o.x * d.x + o.y * d.y - k * rz * d.z,
                                o.x * o.x + o.y * o.y - rz * rz, t0, t1);
      for (int i = 0; i < nr; ++i) {
        const double t = i ? t1 : t0;
        const double z = o.z + t * d.z;
        if (z >= -tol && z <= H + tol && (R1 + k * z) >= -tol)
          consider(sv, q, t, p, 0, group, cond_off, cond_cnt);
      }
    }
    if (fmask & 6) {
      const double invz = frcp(d.z);
#pragma unroll
      for (int f = 1; f <= 2; ++f) {
        if (!((fmask >> f) & 1)) continue;
        const double rc = (f == 1) ? R1 : R2;
        if (!(rc > 0)) continue;
        const double t = (((f == 1) ? 0.0 : H) - o.z) * invz;
        const double px = o.x + t * d.x, py = o.y + t * d.y;
        if (px * px + py * py <= (rc + tol) * (rc + tol))
          consider(sv, q, t, p, f, group, cond_off, cond_cnt);
      }
    }
  } else {  // torus
    if (!(fmask & 1)) return;
    const double R1 = par[0], R2 = par[1];
    const double t0 = -dot(o, d);          // |d| = 1
    const d3 c = o + d * t0;               // closest approach to the centre
    const double bound = (R1 + R2) * 1.0000001 + 1e-9;
    const double h2 = bound * bound - dot(c, c);
    if (!(h2 > 0)) return;                 // misses the bounding sphere
    double s_lo = -fsqrt(h2), s_hi = -s_lo;
    // slab |z| <= R2 (+slack): the torus lies inside it
    const double zs = R2 * 1.0000001 + 1e-9;
    if (d.z != 0) {
      const double iz = frcp(d.z);
      double a = (-zs - c.z) * iz, b = (zs - c.z) * iz;
      if (a > b) { const double tmp = a; a = b; b = tmp; }
      s_lo = fmax(s_lo, a);
      s_hi = fmin(s_hi, b);
    } else if (fabs(c.z) > zs) {
      return;
    }
    if (!(s_hi > s_lo)) return;
    // through the hole: rho^2 is convex in s, its maximum over the clipped
    // segment is at an end point
    {
      const double xa = c.x + s_lo * d.x, ya = c.y + s_lo * d.y;
      const double xb = c.x + s_hi * d.x, yb = c.y + s_hi * d.y;
      const double rin = (R1 - R2) * 0.9999999 - 1e-9;
      if (rin > 0 && xa * xa + ya * ya < rin * rin && xb * xb + yb * yb < rin * rin) return;
    }
    double roots[4];
    const int nr = torus_roots(dot(d, d), 2 * dot(c, d), dot(c, c) + R1 * R1 - R2 * R2,
                               d.x * d.x + d.y * d.y, 2 * (c.x * d.x + c.y * d.y),
                               c.x * c.x + c.y * c.y, R1, s_lo, s_hi, roots);
    for (int i = 0; i < nr; ++i) consider(sv, q, roots[i] + t0, p, 0, group, cond_off, cond_cnt);
  }
}

// outward normal of face `face` of primitive p at local point lp
__device__ __forceinline__ d3 face_normal(int type, cf64 par, int face, d3 lp) {
  if (type == ODW_PRIM_BOX) {
    const double s = (face & 1) ? 1.0 : -1.0;
    const int a = face >> 1;
    return mk(a == 0 ? s : 0.0, a == 1 ? s : 0.0, a == 2 ? s : 0.0);
  }
  if (type == ODW_PRIM_SPHERE) return lp * frsqrt(dot(lp, lp));
  if (type == ODW_PRIM_CYLINDER || type == ODW_PRIM_CONE) {
    if (face == 1) return mk(0, 0, -1);
    if (face == 2) return mk(0, 0, 1);
    double k = 0;
    if (type == ODW_PRIM_CONE) k = (par[1] - par[0]) / par[2];
    const d3 g = mk(lp.x, lp.y, -k * (par[0] + k * lp.z));
    return g * frsqrt(dot(g, g));
  }
  const double f = 1.0 - par[0] * frsqrt(lp.x * lp.x + lp.y * lp.y);
  const d3 g = mk(lp.x * f, lp.y * f, lp.z);
  return g * frsqrt(dot(g, g));
}

// slab test against a global AABB (already enlarged by the tolerance slack).
// A NaN from 0*inf drops out of fmin/fmax, i.e. that axis does not cull.
#define ODW_BVH_STACK 32
template <class P>
__device__ __forceinline__ bool ray_box(P bx, d3 o, d3 inv, double tmax) {
  double t0 = (bx[0] - o.x) * inv.x, t1 = (bx[3] - o.x) * inv.x;
  double lo = fmin(t0, t1), hi = fmax(t0, t1);
  t0 = (bx[1] - o.y) * inv.y; t1 = (bx[4] - o.y) * inv.y;
  lo = fmax(lo, fmin(t0, t1)); hi = fmin(hi, fmax(t0, t1));
  t0 = (bx[2] - o.z) * inv.z; t1 = (bx[5] - o.z) * inv.z;
  lo = fmax(lo, fmin(t0, t1)); hi = fmin(hi, fmax(t0, t1));
  return hi >= fmax(lo, 0.0) && lo < tmax;
}

// findNearestIntersection (ray.py:290-452).  Returns prim (<0: none).
// BVH=false: flat loop, wave-uniform primitive index (scalar loads), each
//            primitive culled by its bounding box first -- the analogue of the
//            reference's shell/face BoundBox culls (ray.py:353-398);
// BVH=true : stack traversal, node stack in LDS (one column per thread).
template <bool BVH>
__device__ __forceinline__ int nearest(const DeviceScene& sc, const SceneView& sv,
                                       const DeviceLimits& lim, d3 start, d3 dn, int medium,
                                       uint64_t mask, double& t_hit, int& face,
                                       int* __restrict__ stack) {
  Query q;
  q.start = start; q.dn = dn; q.tol = lim.dist_tol; q.tmax = lim.max_ray_length + lim.dist_tol;
  q.medium = medium;
  q.any.t = INFINITY; q.any.prim = 0x7fffffff; q.any.face = 0x7fffffff;
  q.oth = q.any;
  const d3 inv = mk(frcp(dn.x), frcp(dn.y), frcp(dn.z));
  if (!BVH) {
    for (int p = 0; p < sc.n_prims; ++p) {
      const int g = sv.prim_i32[4 * p + 1];
      if (!((mask >> g) & 1)) continue;
      // candidates farther than the nearest hit + 2*distTol can never be
      // selected (ray.py:432,440): shrink the search like the reference does
      const double cut = fmin(q.tmax, q.any.t + 2.0 * q.tol);
      if (!ray_box(sv.prim_box + 6 * p, start, inv, cut)) continue;
      intersect_prim(sv, q, p);
    }
  } else {
    cf64 bvh_box = as_const(sc.bvh_box);
    ci32 bvh_link = as_const(sc.bvh_link);
    ci32 bvh_prims = as_const(sc.bvh_prims);
    int sp = 0;
    int node = 0;
    for (;;) {
      const double cut = fmin(q.tmax, q.any.t + 2.0 * q.tol);
      bool descend = ray_box(bvh_box + (size_t)node * 6, start, inv, cut);
      if (descend) {
        const int lk_x = bvh_link[4 * node], lk_y = bvh_link[4 * node + 1], lk_z = bvh_link[4 * node + 2];
        if (lk_x < 0) {  // leaf: ~first, count
          const int first = ~lk_x;
          for (int i = 0; i < lk_y; ++i) {
            const int p = bvh_prims[first + i];
            const int g = sv.prim_i32[4 * p + 1];
            if ((mask >> g) & 1) intersect_prim(sv, q, p);
          }
          descend = false;
        } else {
          // near child first: the left child holds the smaller centroids
          // along the split axis
          const bool fwd = comp(dn, lk_z) >= 0;
          stack[sp * 256] = fwd ? lk_y : lk_x;  // LDS stack, one column per thread
          ++sp;
          node = fwd ? lk_x : lk_y;
        }
      }
      if (!descend) {
        if (sp == 0) break;
        --sp;
        node = stack[sp * 256];
      }
    }
  }
  if (q.any.prim == 0x7fffffff) return -1;
  // hits within 2*distTol of the nearest: prefer one whose group differs
  // from the current medium (ray.py:438-452)
  const Best& sel = (q.oth.prim != 0x7fffffff && q.oth.t < q.any.t + 2.0 * q.tol) ? q.oth : q.any;
  t_hit = sel.t;
  face = sel.face;
  return sel.prim;
}

// ------------------------------------------ mirror / Snell / grating
__device__ __forceinline__ d3 mirror(d3 r, d3 n) { return r - n * (2.0 * dot(r, n)); }

__device__ __forceinline__ d3 snells_law(d3 r, double n1, double n2, d3 n, bool& tir) {
  const d3 c = cross(n, r);
  const double mu = n1 * frcp(n2);
  const double root = 1.0 - mu * mu * dot(c, c);
  if (root < 0) { tir = true; return mirror(r, n); }
  tir = false;
  // n x ((-n) x r) = r (n.n) - n (n.r)
  const d3 perp = cross(n, cross(n * -1.0, r));
  return perp * mu + n * fsqrt(root);
}

__device__ __noinline__ d3 line_grating(d3 ray, double n1, double n2, d3 normal, double wavelength_nm,
                                        int order, double lpm, d3 gdir, bool transmission) {
  const double wl = wavelength_nm / 1000;
  ray = ray * (1.0 / sqrt(dot(ray, ray)));
  const d3 sn = normal * (1.0 / sqrt(dot(normal, normal)));
  const d3 g = gdir * (1.0 / sqrt(dot(gdir, gdir)));
  d3 P = cross(g, sn);
  P = P * (1.0 / sqrt(dot(P, P)));
  d3 D = cross(sn, P);
  D = D * (1.0 / sqrt(dot(D, D)));
  const double mu = n1 / n2, d = 1000 / lpm;
  const double T = (order * wl) / (n1 * d);
  const double V = (mu * dot(ray, sn)) / dot(sn, sn);
  const double W = (mu * mu - 1 + T * T - 2 * mu * T * dot(ray, D)) / dot(sn, sn);
  const double sq = sqrt((2 * V) * (2 * V) - 4 * W);
  const double Q0 = (-2 * V + sq) / 2, Q1 = (-2 * V - sq) / 2;
  const double Q = transmission ? fmin(Q0, Q1) : fmax(Q0, Q1);
  const d3 S = ray * mu - D * T + sn * Q;
  return S * -1.0;
}

// ------------------------------------------------------- recording (K5)
// Active lanes append one 64-B row each: one atomic per wave reserves the
// block, lanes take consecutive slots by popcount prefix of the ballot.
__device__ __forceinline__ void record_hit(const TraceParams& P, uint64_t ray, int group, d3 p, d3 d,
                                           double power, bool entering, uint32_t& n_over,
                                           uint32_t& n_drop) {
  if (P.flags & ODW_TRACE_RECORD_HITS) {
    const uint64_t active = __ballot(1);
    const int lane = __lane_id();
    const int leader = __ffsll((unsigned long long)active) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(P.out.hit_count, (unsigned long long)__popcll(active));
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    const uint64_t slot = (((uint64_t)bhi << 32) | blo) + __popcll(active & ((1ull << lane) - 1ull));
    if (slot < P.out.hit_capacity) {
      double2* row = reinterpret_cast<double2*>(P.out.hits + slot);
      const uint64_t tag = (ray & 0xFFFFFFFFFFFFull) | ((uint64_t)group << 48) | ((uint64_t)entering << 63);
      row[0] = make_double2(p.x, p.y);
      row[1] = make_double2(p.z, d.x);
      row[2] = make_double2(d.y, d.z);
      row[3] = make_double2(power, __longlong_as_double((long long)tag));
    } else {
      ++n_drop;
    }
  }
  if ((P.flags & ODW_TRACE_HISTOGRAM) && P.det.enabled && (P.det.group < 0 || P.det.group == group)) {
#pragma clang fp contract(off)
    const d3 r = p - mk(P.det.origin[0], P.det.origin[1], P.det.origin[2]);
    const double x = dot(r, mk(P.det.ex[0], P.det.ex[1], P.det.ex[2]));
    const double y = dot(r, mk(P.det.ey[0], P.det.ey[1], P.det.ey[2]));
    const double fx = floor((x - P.det.x_lo) * P.det.x_scale);
    const double fy = floor((y - P.det.y_lo) * P.det.y_scale);
    if (fx >= 0 && fx < (double)P.det.nx && fy >= 0 && fy < (double)P.det.ny)
      atomicAdd(P.out.hist + ((size_t)fx * (size_t)P.det.ny + (size_t)fy), 1ull);
    else
      ++n_over;
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------ the kernel
#ifndef ODW_WAVES_PER_SIMD
#define ODW_WAVES_PER_SIMD 2
#endif
template <bool BVH>
__global__ __launch_bounds__(256, ODW_WAVES_PER_SIMD) void odw_trace_kernel(const TraceParams P) {
  extern __shared__ int bvh_stack[];  // ODW_BVH_STACK x 256 ints (BVH variant only)
  const DeviceScene& sc = P.scene;
  const DeviceLimits& lim = P.lim;
  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_box = as_const(sc.prim_box);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  cf64 group_f64 = as_const(sc.group_f64);
  ci32 group_i32 = as_const(sc.group_i32);
  cf64 group_gdir = as_const(sc.group_gdir);
  cu64 seq_mask = as_const(sc.seq_mask);
  uint32_t c_rays = 0, c_hits = 0, c_seg = 0, c_esc = 0, c_died = 0, c_cap = 0, c_over = 0, c_drop = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n_rays; i += stride) {
    const uint64_t ray = P.first_ray + i;
    d3 point, dir;
    double power;
    if (P.ray_origins) {
      point = mk(P.ray_origins[3 * i], P.ray_origins[3 * i + 1], P.ray_origins[3 * i + 2]);
      dir = mk(P.ray_dirs[3 * i], P.ray_dirs[3 * i + 1], P.ray_dirs[3 * i + 2]);
      dir = dir * (1.0 / sqrt(dot(dir, dir)));
      power = P.ray_powers ? P.ray_powers[i] : 1.0;
    } else {
      double up, ut, t, phi;
      ray_uniforms(ray, P.seed, up, ut);
      sample_source(P.source, up, ut, t, phi);
      make_ray(P.source, t, phi, point, dir);
      power = P.source.power;
    }
    // `dir` stays a unit vector: mirror() preserves length, snells_law() and
    // line_grating() return unit vectors for unit input; the reference
    // renormalises every segment (ray.py:377), a no-op up to rounding
    int seq = 0, nint = 0, medium = -1;
    for (;;) {
      if (nint >= lim.max_intersections) { ++c_cap; break; }
      ++nint;
      ++c_seg;
      uint64_t mask = sc.all_mask;
      if (sc.seq_enabled) mask = (seq < sc.seq_len) ? seq_mask[seq] : 0ull;
      mask &= ~sc.ignore_mask;
      double t_hit;
      int face;
      const int prim = nearest<BVH>(sc, sv, lim, point, dir, medium, mask, t_hit, face,
                                    bvh_stack + threadIdx.x);
      if (prim < 0) { ++c_esc; break; }
      cf64 pf = sv.prim_f64 + (size_t)prim * 16;
      ci32 pi = sv.prim_i32 + 4 * prim;
      point = point + dir * t_hit;
      // absorption along the traversed medium (ray.py:120-125, assignment)
      if (medium >= 0) {
        const double L = group_f64[4 * medium + 2];
        if (L == 0) power = 0;
        else if (L < INFINITY) power = exp(-t_hit / L);
      }
      // getNormal (ray.py:455-480): outward normal -> along the travel direction
      d3 n = face_normal(pi[0], pf + 12, face, xf_point(pf, point));
      if (pi[2] & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
      n = xf_vec_t(pf, n);
      const bool entering = dot(dir, n) < 0;
      if (entering) n = n * -1.0;
      const int g = pi[1];
      const int gtype = group_i32[4 * g];
      if (group_i32[4 * g + 1]) {
        ++c_hits;
        record_hit(P, ray, g, point, dir, power, entering, c_over, c_drop);
      }
      if (gtype == ODW_OPT_MIRROR) {
        dir = mirror(dir, n);
        power *= group_f64[4 * g + 1];
        ++seq;
      } else if (gtype == ODW_OPT_LENS) {
        const double n1 = (medium >= 0) ? group_f64[4 * medium] : 1.0;
        double n2 = 1.0;
        if (entering) { medium = g; n2 = group_f64[4 * g]; }
        bool tir;
        dir = snells_law(dir, n1, n2, n, tir);
        if (!entering && !tir && medium == g) { medium = -1; ++seq; }
      } else if (gtype == ODW_OPT_ABSORBER) {
        power = 0;
        ++seq;
      } else if (gtype == ODW_OPT_VACUUM) {
        ++seq;
      } else {  // grating (ray.py:216-268)
        const d3 gd = mk(group_gdir[3 * g], group_gdir[3 * g + 1], group_gdir[3 * g + 2]);
        const double lpm = group_f64[4 * g + 3];
        const int order = group_i32[4 * g + 3];
        if (group_i32[4 * g + 2] == 0) {
          if (entering) {
            const double nn = (medium >= 0) ? group_f64[4 * medium] : 1.0;
            dir = line_grating(dir, nn, nn, n, P.source.wavelength, order, lpm, gd, false);
            ++seq;
          }
        } else if (entering) {
          if (medium >= 0) { ++c_died; break; }
          medium = g;
          dir = line_grating(dir, 1.0, group_f64[4 * g], n, P.source.wavelength, order, lpm, gd, true);
        } else {
          const double n1 = (medium >= 0) ? group_f64[4 * medium] : 1.0;
          bool tir;
          dir = snells_law(dir, n1, 1.0, n, tir);
          if (!tir) { medium = -1; ++seq; }
        }
      }
      if (power < lim.power_tol) { ++c_died; break; }
    }
    ++c_rays;
  }
  // counters: wave reduction, one atomic per wave and counter
  uint32_t v[ODW_CNT_COUNT] = {c_rays, c_hits, c_seg, c_esc, c_died, c_cap, c_over, c_drop};
#pragma unroll
  for (int k = 0; k < ODW_CNT_COUNT; ++k) {
    const uint32_t s = wave_sum(v[k]);
    if (__lane_id() == 0 && s) atomicAdd(P.out.counters + k, (unsigned long long)s);
  }
}

// sampler only: theta-or-radius and phi per ray (diagnostics / parity tests)
__global__ __launch_bounds__(256) void odw_sample_kernel(const DeviceSource s, uint64_t first,
                                                         uint64_t n, uint64_t seed,
                                                         double* __restrict__ t_out,
                                                         double* __restrict__ phi_out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double up, ut, t, phi;
    ray_uniforms(first + i, seed, up, ut);
    sample_source(s, up, ut, t, phi);
    t_out[i] = t;
    phi_out[i] = phi;
  }
}

}  // namespace odw
