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
//                 prefix; slots reserved per wave, 512 at a time for big
//                 lists) and u64 histogram scatter
//                 (optical_group.py:206-209 -> results_store.py:641-648)
// plus, in the kernel variants that need them: stochastic surfaces (scatter),
// triangle primitives (BVH kernels), and the surface-source emission kernel.
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
// a uniform pointer the optimiser cannot see through: loads behind it stay
// where they are written instead of being hoisted out of the ray loop
template <class T>
__device__ __forceinline__ const T* opaque(const T* p) {
  asm volatile("" : "+s"(p));
  return p;
}
typedef const DeviceSource ODW_CONST* csource;
typedef const DeviceDetector ODW_CONST* cdetector;

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
// one Newton step (~2^-50): for the inverse direction of the box tests, whose boxes carry 2 distTol of slack
__device__ __forceinline__ double frcp1(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
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
    // (one 32 x 32 -> 64 bit multiply per product: v_mad_u64_u32, instead of a high and a low one --
    //  integer multiplies issue at a quarter of the rate)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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

struct TableView {
  const double* phi_tab; const double* t_tab; const int32_t* t_guide;
  int n_phi_knots, n_t_knots, n_t_rows, n_guide;
  const int32_t* phi_guide; int n_phi_guide;      // optional (null: plain binary search)
};
__device__ __forceinline__ void sample_tables(const TableView& s, double u_phi, double u_t,
                                              double& t_out, double& phi_out) {
#pragma clang fp contract(off)
  int plo = 0, phi_hi = s.n_phi_knots - 1;
  if (s.phi_guide) {                     // bracket from the guide: two independent loads instead of a chain of ~7
    const int k = (int)(u_phi * (double)s.n_phi_guide);
    plo = s.phi_guide[k];
    phi_hi = min(s.phi_guide[k + 1] + 1, s.n_phi_knots - 1);
  }
  const double phi = inv_cdf(s.phi_tab, plo, phi_hi, u_phi);
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

__device__ __forceinline__ void sample_source(csource sp_, double u_phi, double u_t,
                                              double& t_out, double& phi_out) {
  TableView s;
  s.phi_tab = sp_->phi_tab; s.t_tab = sp_->t_tab; s.t_guide = sp_->t_guide;
  s.n_phi_knots = sp_->n_phi_knots; s.n_t_knots = sp_->n_t_knots; s.n_t_rows = sp_->n_t_rows; s.n_guide = sp_->n_guide;
  s.phi_guide = sp_->phi_guide; s.n_phi_guide = sp_->n_phi_guide;
  sample_tables(s, u_phi, u_t, t_out, phi_out);
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

// ---- scene-compiled kernels (odw_spec.h) -----------------------------------
// A SPEC class carries the STRUCTURE of one baked scene as compile-time constants: primitive types,
// groups, flags, trimming lists, which entries of every global->local frame are exactly zero.  The
// primitive loop of the flat kernel is then unrolled over it: type dispatch, face masks and condition
// lists fold away, frame products skip their zero terms, and every table read has a constant offset
// (no dependent scalar loads, no scalar branches).  All float64 VALUES (frames, parameters, boxes,
// optical constants) are still read from the tables, so one compiled kernel serves every scene of the
// same structure (a radius sweep compiles once).  NoSpec = the generic kernels.
struct NoSpec {
  static constexpr bool enabled = false;
  static constexpr int N = 0;
  static constexpr unsigned long long xf(int) { return 0xfffull; }
  static constexpr int type(int) { return 0; }
  static constexpr int cond(int) { return 0; }
  static constexpr bool isolated() { return true; }
};
// Frame products of a compiled scene.  XF = the pattern of a primitive's 12 frame entries: bits 0-11
// entry != 0, bits 12-23 entry == +1, bits 24-35 entry == -1.  A term whose coefficient is exactly zero
// is left out, a coefficient of +-1 becomes an addition / subtraction; the order of the remaining
// operations is the generic one (second product first, then the first and third by fma, then the
// translation), so the results are those of xf_point / xf_vec / xf_vec_t bit for bit (apart from the
// sign of a zero).
template <unsigned long long XF, int IA, int IB, int IC, int IT, class P>
__device__ __forceinline__ double xf_comb(P m, double a, double b, double c) {
  double acc = 0.0;
  bool have = false;
#define ODW_XF_TERM(I, V)                                                                            \
  if constexpr (((XF >> (I)) & 1ull) != 0) {                                                         \
    if constexpr (((XF >> (12 + (I))) & 1ull) != 0) acc = have ? acc + (V) : (V);                    \
    else if constexpr (((XF >> (24 + (I))) & 1ull) != 0) acc = have ? acc - (V) : -(V);              \
    else acc = have ? fma(m[I], (V), acc) : m[I] * (V);                                              \
    have = true;                                                                                     \
  }
  ODW_XF_TERM(IB, b)
  ODW_XF_TERM(IA, a)
  ODW_XF_TERM(IC, c)
#undef ODW_XF_TERM
  if constexpr (IT >= 0) {
    if constexpr (((XF >> (IT < 0 ? 0 : IT)) & 1ull) != 0) acc = have ? m[IT < 0 ? 0 : IT] + acc : m[IT < 0 ? 0 : IT];
  }
  return acc;
}
template <unsigned long long XF, class P>
__device__ __forceinline__ d3 xf_point_nz(P m, d3 p) {
  return mk(xf_comb<XF, 0, 1, 2, 3>(m, p.x, p.y, p.z), xf_comb<XF, 4, 5, 6, 7>(m, p.x, p.y, p.z),
            xf_comb<XF, 8, 9, 10, 11>(m, p.x, p.y, p.z));
}
template <unsigned long long XF, class P>
__device__ __forceinline__ d3 xf_vec_nz(P m, d3 v) {
  return mk(xf_comb<XF, 0, 1, 2, -1>(m, v.x, v.y, v.z), xf_comb<XF, 4, 5, 6, -1>(m, v.x, v.y, v.z),
            xf_comb<XF, 8, 9, 10, -1>(m, v.x, v.y, v.z));
}
template <unsigned long long XF, class P>
__device__ __forceinline__ d3 xf_vec_t_nz(P m, d3 v) {  // R^T v
  return mk(xf_comb<XF, 0, 4, 8, -1>(m, v.x, v.y, v.z), xf_comb<XF, 1, 5, 9, -1>(m, v.x, v.y, v.z),
            xf_comb<XF, 2, 6, 10, -1>(m, v.x, v.y, v.z));
}

// sin and cos for |x| <~ 64 (source angles are domain-limited): one Cody-Waite
// reduction step by pi/2 with fma, fdlibm kernel polynomials; <= 1 ulp of 1.0
// absolute error (checked against libm on 4e6 points).  Replaces ocml's
// sincos(double), whose large-argument path costs ~20 extra VGPRs.
__device__ __forceinline__ void sincos_bounded(double x, double& s, double& c) {
  const double k = rint(x * 0.63661977236758134308);
  double r = fma(-k, 1.5707963267948966, x);
  r = fma(-k, 6.123233995736766e-17, r);
  const double z = r * r;
  const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sn = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double cs = 1.0 - (0.5 * z - z * pc);
  const int q = (int)k & 3;
  const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

// PointSourceProxy._makeRay (point_source.py:411-460)
__device__ __forceinline__ void make_ray(csource s, double t_or_r, double phi,
                                         d3& origin, d3& dir) {
  d3 ldir, lorg;
  double sp, cp;
  sincos_bounded(phi, sp, cp);
  if (s->finite_focal) {
    double st, ct;
    sincos_bounded(t_or_r, st, ct);
    ldir = mk(st * sp, -st * cp, ct);
    lorg = (mk(0, 0, 1) - ldir) * s->focal_length;
  } else {
    ldir = mk(0, 0, 1);
    lorg = mk(t_or_r * cp, -t_or_r * sp, 0.0);
  }
  // (both vectors are unit vectors up to rounding already: the two normalisations of the reference move them
  //  by an ulp or two, frsqrt -- <= 2 ulp -- does the same without the IEEE square root and division)
  const d3 ln = ldir * frsqrt(dot(ldir, ldir));
  const d3 p1 = xf_point(s->m, lorg);
  const d3 p2 = xf_point(s->m, lorg + ln);
  const d3 d = p2 - p1;
  origin = p1;
  dir = d * frsqrt(dot(d, d));
}

// ----------------------------------------------------------- primitives
// PARAB: paraboloids are known to the BVH and grid kernels only (the flat kernel of the small
// benchmark scenes is sensitive to every instruction in its loop: their branch cost it 1.1 %)
template <bool PARAB, class PP>
__device__ __forceinline__ double prim_sdist(int type, PP par, d3 p) {
  if (PARAB && type == ODW_PRIM_PARABOLOID) {
    // x^2 + y^2 - 4 f z over the length of its gradient: the distance to first order
    const double r2 = p.x * p.x + p.y * p.y;
    const double lat = (r2 - 4.0 * par[0] * p.z) * 0.5 * frsqrt(r2 + 4.0 * par[0] * par[0]);
    return fmax(lat, p.z - par[1]);
  }
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

// (s >= 0 ? m : -m) for m >= 0 without a compare and two selects: the sign bit of s + 0.0 (which turns -0 into +0, so
// that s = -0 counts as s >= 0, as in the comparison) copied onto m -- one add, one bit-field insert
__device__ __forceinline__ double signed_like(double m, double s) { return __builtin_copysign(m, s + 0.0); }

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
  const double q = -(bh + signed_like(sq, bh));
  const double r0 = q * frcp(a);
  const double r1 = (q != 0) ? c * frcp(q) : r0;
  t0 = fmin(r0, r1); t1 = fmax(r0, r1);
  return 2;
}
// the same for a == 1 (unit direction in a rigid frame)
__device__ __forceinline__ int quad_roots_unit(double bh, double c, double& t0, double& t1) {
  const double disc = bh * bh - c;
  if (!(disc >= 0)) return 0;
  const double sq = fsqrt(disc);
  const double q = -(bh + signed_like(sq, bh));
  const double r1 = (q != 0) ? c * frcp(q) : q;
  t0 = fmin(q, r1); t1 = fmax(q, r1);
  return 2;
}

// Diagnostic builds (-DODW_DOUBLE=k, through ODW_SPEC_OPTS for the compiled kernels): piece k of the ray loop is
// computed TWICE, the second time behind an operand the compiler cannot see through, and the two outcomes are merged so
// that the results stay what they are -- the launch's extra time is what that piece costs as it runs (all lanes, all
// waves, cache and scheduling effects included).  1 box tests, 2 sphere roots, 3 cylinder / cone side and caps, 4 box
// faces, 5 trimming tests, 6 normal at the hit, 7 mirror / Snell, 8 ray generation, 9 inverse direction of a segment,
// 12 sphere (whole candidate pass); grid kernel (library builds, scripts/build_variant.py): 21 the walk's sphere test,
// 23 the exact roots of a resolved cell.  scripts/gpu_double_profile.sh runs the compiled kernel's.
#ifndef ODW_DOUBLE
#define ODW_DOUBLE 0
#endif
__device__ __forceinline__ double opq(double x) { asm volatile("" : "+v"(x)); return x; }

// A/B (round 4): conditions without short-circuit evaluation (compares and scalar ands in a row instead of nests of
// exec-mask branches with their copies) -- bit 0 box faces, 1 the pick among a primitive's candidates, 2 cylinder side
// and caps, 3 better()
#ifndef ODW_SPEC_KEY
#define ODW_SPEC_KEY 1           // (A/B: 0 = primitive and face as two members of a running minimum)
#endif
#ifndef ODW_SPEC_LEANCONS
#define ODW_SPEC_LEANCONS 1       // (A/B: 0 = every candidate slot of a trimmed primitive goes through consider_spec)
#endif
#ifndef ODW_SPEC_CUT
#define ODW_SPEC_CUT 1            // (A/B: 0 = the end of the search computed anew before every box test)
#endif
#ifndef ODW_OTH_LAZY
#define ODW_OTH_LAZY 1           // (A/B: 0 = the second running minimum kept for every candidate)
#endif
#ifndef ODW_FLAT_NOBRANCH
#define ODW_FLAT_NOBRANCH 15
#endif
// ------------------------------------------------------------------------
struct Best {
  double t;
  int prim, face;
};
__device__ __forceinline__ bool better(double t, int p, int f, const Best& b) {
#if ODW_FLAT_NOBRANCH & 8
  return (bool)((int)(t < b.t) | ((int)(t == b.t) & ((int)(p < b.prim) | ((int)(p == b.prim) & (int)(f < b.face)))));
#else
  if (t != b.t) return t < b.t;
  if (p != b.prim) return p < b.prim;
  return f < b.face;
#endif
}

struct Query {
  d3 start, dn;       // global ray (unit direction)
  double tol, tmax;   // distTol, maxRayLength + distTol
  int medium;
  bool in_medium;     // (compiled kernels) some lane of the wave is inside a medium: the second running minimum is kept
  double cut;         // (compiled kernels, ODW_SPEC_CUT) min(tmax, nearest + 2 tol): where the search ends, renewed when the nearest changes
  Best any, oth;
};

struct SceneView {    // constant-address-space views of the scene tables
  cf64 prim_f64, prim_hdr;
  ci32 prim_i32, cond_i32;
};

// One candidate intersection of primitive p: range test, then -- only if it
// would replace a running minimum -- the trim by the other operands of a
// boolean (cond list), then bookkeeping of the two running minima (nearest of
// all / nearest whose group is not the current medium).
template <bool PARAB = true>
__device__ __forceinline__ void consider(const SceneView& sv, Query& q, double t, int p, int face,
                                         int group, int cond_off, int cond_cnt) {
  if (!(t > q.tol && t < q.tmax)) return;
  const bool cand_any = better(t, p, face, q.any);
  // (in vacuum the second running minimum would be the first all along: see consider_spec)
  const bool cand_oth = (!ODW_OTH_LAZY || q.medium >= 0) && (group != q.medium) && better(t, p, face, q.oth);
  if (!cand_any && !cand_oth) return;
  if (cond_cnt) {
    const d3 gp = q.start + q.dn * t;
    for (int c = cond_off; c < cond_off + cond_cnt; ++c) {
      const int cw = sv.cond_i32[c];
      const int qp = cw & 0x7fffffff;
      cf64 pf = sv.prim_f64 + (size_t)qp * 16;
      const double sd = prim_sdist<PARAB>(sv.prim_i32[4 * qp], pf + 12, xf_point(pf, gp));
      if (cw < 0) { if (sd > q.tol) return; }     // must be inside
      else { if (sd < -q.tol) return; }           // must be outside
    }
  }
  if (cand_any) { q.any.t = t; q.any.prim = p; q.any.face = face; }
  if (cand_oth) { q.oth.t = t; q.oth.prim = p; q.oth.face = face; }
}

// the same for primitive PI of a compiled scene: group and trimming list are constants, the list is
// unrolled, every operand's frame product skips its zero terms
template <bool PARAB, class SPEC, int C, int END>
__device__ __forceinline__ bool trim_ok(const SceneView& sv, const Query& q, d3 gp) {
  if constexpr (C >= END) {
    return true;
  } else {
    constexpr int cw = SPEC::cond(C);
    constexpr int qp = cw & 0x7fffffff;
    cf64 pf = sv.prim_f64 + (size_t)qp * 16;
    const double sd = prim_sdist<PARAB>(SPEC::type(qp), pf + 12, xf_point_nz<SPEC::xf(qp)>(pf, gp));
    if (cw < 0) { if (sd > q.tol) return false; }     // must be inside
    else { if (sd < -q.tol) return false; }           // must be outside
    return trim_ok<PARAB, SPEC, C + 1, END>(sv, q, gp);
  }
}
template <bool PARAB, class SPEC, int PI>
__device__ __forceinline__ void consider_spec(const SceneView& sv, Query& q, double t, int face) {
  if (!(t > q.tol && t < q.tmax)) return;
#if ODW_SPEC_KEY
  // (compiled kernels keep (primitive, face) as ONE word, primitive << 8 | face, in the `face` member: the order of the
  //  pairs is the order of the words -- one integer comparison per running minimum instead of three, one select less
  //  per update; nearest() takes the word apart once per segment)
  const int key = (PI << 8) | face;
#define ODW_BETTER(B) ((bool)((int)(t < (B).t) | ((int)(t == (B).t) & (int)(key < (B).face))))
#else
  const int key = face;
#define ODW_BETTER(B) better(t, PI, face, B)
#endif
  const bool cand_any = ODW_BETTER(q.any);
  // (a ray in vacuum: every candidate's group differs from its medium, `oth` would be `any` all along and the rule at
  //  the end of nearest() returns `any` either way -- a wave whose lanes are all in vacuum leaves the second minimum
  //  alone: a uniform branch around its comparison and its selects)
  bool cand_oth = false;
  if (!ODW_OTH_LAZY || q.in_medium) cand_oth = (SPEC::group(PI) != q.medium) && (!ODW_OTH_LAZY || q.medium >= 0) && ODW_BETTER(q.oth);
#undef ODW_BETTER
  if (!cand_any && !cand_oth) return;
  if constexpr (SPEC::cond_cnt(PI) > 0) {
    if (!trim_ok<PARAB, SPEC, SPEC::cond_off(PI), SPEC::cond_off(PI) + SPEC::cond_cnt(PI)>(sv, q, q.start + q.dn * t)) return;
#if ODW_DOUBLE == 5
    if (!trim_ok<PARAB, SPEC, SPEC::cond_off(PI), SPEC::cond_off(PI) + SPEC::cond_cnt(PI)>(sv, q, q.start + q.dn * opq(t))) return;
#endif
  }
#if ODW_SPEC_KEY
  if (cand_any) { q.any.t = t; q.any.face = key; if (ODW_SPEC_CUT) q.cut = fmin(q.tmax, t + 2.0 * q.tol); }
  if (cand_oth) { q.oth.t = t; q.oth.face = key; }
#else
  if (cand_any) { q.any.t = t; q.any.prim = PI; q.any.face = face; }
  if (cand_oth) { q.oth.t = t; q.oth.prim = PI; q.oth.face = face; }
#endif
}

// up to four candidate (t, face) pairs of one primitive, kept in registers
struct Cands {
  double t0, t1, t2, t3;
  int f0, f1, f2, f3;
};
__device__ __forceinline__ void cand_push(Cands& c, int& n, double t, int f) {
  // n is compile-time after unrolling at every call site
  if (n == 0) { c.t0 = t; c.f0 = f; }
  else if (n == 1) { c.t1 = t; c.f1 = f; }
  else if (n == 2) { c.t2 = t; c.f2 = f; }
  else { c.t3 = t; c.f3 = f; }
  ++n;
}
// keep the two smallest (t, face) of a stream (box: entry and exit face)
__device__ __forceinline__ void cand_min2(Cands& c, double t, int f) {
  const bool lt0 = t < c.t0 || (t == c.t0 && f < c.f0);
  const bool lt1 = t < c.t1 || (t == c.t1 && f < c.f1);
  c.t1 = lt0 ? c.t0 : (lt1 ? t : c.t1);
  c.f1 = lt0 ? c.f0 : (lt1 ? f : c.f1);
  c.t0 = lt0 ? t : c.t0;
  c.f0 = lt0 ? f : c.f0;
}

#ifndef ODW_CYL_SIDE_SKIP
#define ODW_CYL_SIDE_SKIP 1      // (A/B: 0 = every cylinder side by its quadratic)
#endif
#ifndef ODW_TORUS_PLAIN
#define ODW_TORUS_PLAIN 96     // plain distance steps before the curvature bound joins in
#endif
// every face of primitive p against the ray: untrimmed analytic surface,
// natural face bounds with tolerance (ray.py:411-426).  The candidates are
// collected first and judged by ONE copy of consider() (code size: the hot
// loop must stay inside the instruction cache).
// SPEC / PI: primitive PI of a compiled scene -- the caller passes its constants as p, type, group,
// flags, cond_word, so everything that depends on them folds
template <bool PARAB = true, class SPEC = NoSpec, int PI = 0>
__device__ __forceinline__ void intersect_prim(const SceneView& sv, Query& q, int p, int type, int group,
                                               int flags, int cond_word) {
  cf64 pf = sv.prim_f64 + (size_t)p * 16;
  const int cond_off = cond_word & 0xffffff, cond_cnt = (cond_word >> 24) & 0xff;
  const int fmask = (flags >> ODW_FACEMASK_SHIFT) & 0xff;
  cf64 par = pf + 12;
  const double tol = q.tol;
  Cands c;
  c.t0 = c.t1 = c.t2 = c.t3 = INFINITY;
  c.f0 = c.f1 = c.f2 = c.f3 = 0;
  if (type == ODW_PRIM_SPHERE) {
    // a sphere needs no frame: par[1..3] = its centre in global coordinates (filled in by the host)
    if (fmask & 1) {
      const d3 oc = q.start - mk(par[1], par[2], par[3]);
      double t0, t1;
      if (quad_roots_unit(dot(oc, q.dn), dot(oc, oc) - par[0] * par[0], t0, t1) == 2) {
        c.t0 = t0;
        c.t1 = t1;
      }
#if ODW_DOUBLE == 2
      {
        const d3 oc2 = mk(opq(oc.x), oc.y, oc.z);
        double u0 = INFINITY, u1 = INFINITY;
        if (quad_roots_unit(dot(oc2, q.dn), dot(oc2, oc2) - par[0] * par[0], u0, u1) == 2) { c.t0 = u0 == c.t0 ? c.t0 : u0; c.t1 = u1 == c.t1 ? c.t1 : u1; }
      }
#endif
    }
  } else {
  d3 o, d;
  if constexpr (SPEC::enabled) {
    o = xf_point_nz<SPEC::xf(PI)>(pf, q.start);
    d = xf_vec_nz<SPEC::xf(PI)>(pf, q.dn);
  } else {
    o = xf_point(pf, q.start);
    d = xf_vec(pf, q.dn);
  }
  if (type == ODW_PRIM_BOX) {
    // Slab form of the six plane tests.  Per axis the ray meets the low/high
    // plane at tn <= tf; a face hit is valid when the other two coordinates lie
    // inside the face rectangle (+- tol).  All faces of a primitive share its
    // group, so without trimming conditions only the nearest valid hit can be
    // selected: the three entry faces are tried first, the three exit faces
    // only if no entry face qualifies (ray starts inside / on the box) or
    // the ray misses the exact box (last entry plane behind the first exit
    // plane): passing an edge within the tolerance it can meet the widened
    // rectangle of an exit face before that of an entry face.
    const double ix = frcp(d.x), iy = frcp(d.y), iz = frcp(d.z);
    const double ax = -o.x * ix, ay = -o.y * iy, az = -o.z * iz;            // plane at 0
    const double bx = fma(par[0], ix, ax), by = fma(par[1], iy, ay), bz = fma(par[2], iz, az);
    const bool px = d.x > 0, py = d.y > 0, pz = d.z > 0;
    // per axis the plane met first / last (the low face first when moving in + direction): the smaller / larger of
    // the two distances -- one v_min / v_max each instead of a select (two v_cndmask behind the compare) each
    const double nx_ = fmin(ax, bx), ny_ = fmin(ay, by), nz_ = fmin(az, bz);
    const double fx_ = fmax(ax, bx), fy_ = fmax(ay, by), fz_ = fmax(az, bz);
    if (cond_cnt) {
      // Trimmed box (operand of a boolean): any face can be rejected by its trim, so every valid
      // face has to reach consider() -- within the tolerance of an edge the ray meets the widened
      // rectangles of two entry (or two exit) faces.  Slots 0/2: the two nearest entry faces,
      // 1/3: the two nearest exit faces (a third is possible only within distTol of a corner).
#define ODW_BOX_FACE2(T, FACE, P1, D1, S1, P2, D2, S2, TA, FA, TB, FB)                          \
      {                                                                                         \
        const double t_ = (T);                                                                  \
        const double u_ = fma(t_, D1, P1), v_ = fma(t_, D2, P2);                                \
        const bool ok_ = ((fmask >> (FACE)) & 1) && t_ > tol && u_ >= -tol && u_ <= (S1) + tol && \
                         v_ >= -tol && v_ <= (S2) + tol;                                        \
        const bool a_ = ok_ && t_ < TA;                                                         \
        const bool b_ = ok_ && !a_ && t_ < TB;                                                  \
        TB = a_ ? TA : (b_ ? t_ : TB);                                                          \
        FB = a_ ? FA : (b_ ? (FACE) : FB);                                                      \
        TA = a_ ? t_ : TA;                                                                      \
        FA = a_ ? (FACE) : FA;                                                                  \
      }
      ODW_BOX_FACE2(nx_, px ? 0 : 1, o.y, d.y, par[1], o.z, d.z, par[2], c.t0, c.f0, c.t2, c.f2);
      ODW_BOX_FACE2(ny_, py ? 2 : 3, o.z, d.z, par[2], o.x, d.x, par[0], c.t0, c.f0, c.t2, c.f2);
      ODW_BOX_FACE2(nz_, pz ? 4 : 5, o.x, d.x, par[0], o.y, d.y, par[1], c.t0, c.f0, c.t2, c.f2);
      ODW_BOX_FACE2(fx_, px ? 1 : 0, o.y, d.y, par[1], o.z, d.z, par[2], c.t1, c.f1, c.t3, c.f3);
      ODW_BOX_FACE2(fy_, py ? 3 : 2, o.z, d.z, par[2], o.x, d.x, par[0], c.t1, c.f1, c.t3, c.f3);
      ODW_BOX_FACE2(fz_, pz ? 5 : 4, o.x, d.x, par[0], o.y, d.y, par[1], c.t1, c.f1, c.t3, c.f3);
#undef ODW_BOX_FACE2
    } else {
    double bt = INFINITY;
    int bf = 0;
    // (ODW_FLAT_NOBRANCH: the six conditions of a face as one expression without short-circuit evaluation -- six
    //  compares and five scalar ands in a row instead of a nest of exec-mask branches with their copies)
#if ODW_FLAT_NOBRANCH & 1
#define ODW_BOX_OK(A, B, C, D, E, F) (bool)((int)(A) & (int)(B) & (int)(C) & (int)(D) & (int)(E) & (int)(F))
#else
#define ODW_BOX_OK(A, B, C, D, E, F) ((A) && (B) && (C) && (D) && (E) && (F))
#endif
#define ODW_BOX_FACE(T, FACE, P1, D1, S1, P2, D2, S2)                                         \
    {                                                                                         \
      const double t_ = (T);                                                                  \
      const double u_ = fma(t_, D1, P1), v_ = fma(t_, D2, P2);                                \
      const bool ok_ = ODW_BOX_OK(((fmask >> (FACE)) & 1) != 0, t_ > tol, u_ >= -tol, u_ <= (S1) + tol, \
                                  v_ >= -tol, v_ <= (S2) + tol);                              \
      if (ok_ && (t_ < bt || (t_ == bt && (FACE) < bf))) { bt = t_; bf = (FACE); }            \
    }
    // entry faces: low face when moving in +axis direction
    ODW_BOX_FACE(nx_, px ? 0 : 1, o.y, d.y, par[1], o.z, d.z, par[2]);
    ODW_BOX_FACE(ny_, py ? 2 : 3, o.z, d.z, par[2], o.x, d.x, par[0]);
    ODW_BOX_FACE(nz_, pz ? 4 : 5, o.x, d.x, par[0], o.y, d.y, par[1]);
    c.t0 = bt;
    c.f0 = bf;
    const double t_in = fmax(fmax(nx_, ny_), nz_);
    const double t_out = fmin(fmin(fx_, fy_), fz_);
    if (!(bt < INFINITY) || !(t_in < t_out)) {
      bt = INFINITY;
      bf = 0;
      ODW_BOX_FACE(fx_, px ? 1 : 0, o.y, d.y, par[1], o.z, d.z, par[2]);
      ODW_BOX_FACE(fy_, py ? 3 : 2, o.z, d.z, par[2], o.x, d.x, par[0]);
      ODW_BOX_FACE(fz_, pz ? 5 : 4, o.x, d.x, par[0], o.y, d.y, par[1]);
      c.t1 = bt;
      c.f1 = bf;
    }
#undef ODW_BOX_FACE
#undef ODW_BOX_OK
    }
  } else if (type == ODW_PRIM_CYLINDER || type == ODW_PRIM_CONE || (PARAB && type == ODW_PRIM_PARABOLOID)) {
    // one quadric template: cylinder / cone x^2 + y^2 = (R1 + k z)^2, paraboloid x^2 + y^2 = 4 f z
    // (R1 = k = 0 and a term linear in z; its only cap is the one at z = H, face 2)
    const bool parab = PARAB && type == ODW_PRIM_PARABOLOID;
    const double R1 = parab ? 0.0 : par[0];
    const double R2 = (type == ODW_PRIM_CYLINDER) ? par[0] : (parab ? par[2] : par[1]);
    const double H = (type == ODW_PRIM_CONE) ? par[2] : par[1];
    const double k = (type == ODW_PRIM_CONE) ? (R2 - R1) / H : 0.0;
    const double f2 = parab ? 2.0 * par[0] : 0.0;
    // A cylinder's side without its quadratic (flat kernels): the squared distance from the axis is convex along the
    // ray, so a ray that is inside the radius where it crosses the planes z = -tol and z = H + tol is inside in between
    // -- no root of the side can pass the z test below.  A beam through the inside of a lens is that case for every
    // lane of a wave: two plane distances and two squared radii (14 instructions) instead of discriminant, square root,
    // reciprocal and roots (~55).  The margin (1e-9 of R^2: the roots then lie >= 5e-10 R beyond the planes) is far above
    // the roots' rounding; anything closer, a ray parallel to the caps (NaN), cones and paraboloids take the quadratic.
    constexpr bool side_skip = ODW_CYL_SIDE_SKIP && (SPEC::enabled || !PARAB);
    bool side = (fmask & 1) != 0;
    double invz = 0.0;
    if ((fmask & 6) || (side_skip && type == ODW_PRIM_CYLINDER && side)) invz = frcp(d.z);
    if (side_skip && type == ODW_PRIM_CYLINDER && side) {
      const double tl = (-tol - o.z) * invz, th = (H + tol - o.z) * invz;
      const double xl = fma(tl, d.x, o.x), yl = fma(tl, d.y, o.y), xh = fma(th, d.x, o.x), yh = fma(th, d.y, o.y);
      const double lim = R1 * R1 * (1.0 - 1e-9);
      side = !(fma(xl, xl, yl * yl) < lim && fma(xh, xh, yh * yh) < lim);
    }
    if (side) {
      const double rz = R1 + k * o.z;
      double t0 = INFINITY, t1 = INFINITY;
      const int nr = quad_roots(d.x * d.x + d.y * d.y - k * k * d.z * d.z,
                                o.x * d.x + o.y * d.y - k * rz * d.z - f2 * d.z,
                                o.x * o.x + o.y * o.y - rz * rz - 2.0 * f2 * o.z, t0, t1);
      const double z0 = o.z + t0 * d.z, z1 = o.z + t1 * d.z;
#if ODW_FLAT_NOBRANCH & 4
      c.t0 = (bool)((int)(nr >= 1) & (int)(z0 >= -tol) & (int)(z0 <= H + tol) & (int)((R1 + k * z0) >= -tol)) ? t0 : c.t0;
      c.t1 = (bool)((int)(nr == 2) & (int)(z1 >= -tol) & (int)(z1 <= H + tol) & (int)((R1 + k * z1) >= -tol)) ? t1 : c.t1;
#else
      if (nr >= 1 && z0 >= -tol && z0 <= H + tol && (R1 + k * z0) >= -tol) c.t0 = t0;
      if (nr == 2 && z1 >= -tol && z1 <= H + tol && (R1 + k * z1) >= -tol) c.t1 = t1;
#endif
    }
    if (fmask & 6) {
      const double ta = (0.0 - o.z) * invz, tb = (H - o.z) * invz;
      const double xa = o.x + ta * d.x, ya = o.y + ta * d.y;
      const double xb = o.x + tb * d.x, yb = o.y + tb * d.y;
#if ODW_FLAT_NOBRANCH & 4
      { const bool w_ = (bool)((int)((fmask & 2) != 0) & (int)(R1 > 0) & (int)(xa * xa + ya * ya <= (R1 + tol) * (R1 + tol))); c.t2 = w_ ? ta : c.t2; c.f2 = w_ ? 1 : c.f2; }
      { const bool w_ = (bool)((int)((fmask & 4) != 0) & (int)(R2 > 0) & (int)(xb * xb + yb * yb <= (R2 + tol) * (R2 + tol))); c.t3 = w_ ? tb : c.t3; c.f3 = w_ ? 2 : c.f3; }
#else
      if ((fmask & 2) && R1 > 0 && xa * xa + ya * ya <= (R1 + tol) * (R1 + tol)) { c.t2 = ta; c.f2 = 1; }
      if ((fmask & 4) && R2 > 0 && xb * xb + yb * yb <= (R2 + tol) * (R2 + tol)) { c.t3 = tb; c.f3 = 2; }
#endif
    }
  } else {  // torus
    if (!(fmask & 1)) return;
    const double R1 = par[0], R2 = par[1];
    const double t0 = -dot(o, d);          // |d| = 1
    const d3 cc = o + d * t0;              // closest approach to the centre
    const double bound = (R1 + R2) * 1.0000001 + 1e-9;
    const double h2 = bound * bound - dot(cc, cc);
    if (!(h2 > 0)) return;                 // misses the bounding sphere
    double s_lo = -fsqrt(h2), s_hi = -s_lo;
    // slab |z| <= R2 (+slack): the torus lies inside it
    const double zs = R2 * 1.0000001 + 1e-9;
    if (d.z != 0) {
      const double iz = frcp(d.z);
      double a = (-zs - cc.z) * iz, b = (zs - cc.z) * iz;
      if (a > b) { const double tmp = a; a = b; b = tmp; }
      s_lo = fmax(s_lo, a);
      s_hi = fmin(s_hi, b);
    } else if (fabs(cc.z) > zs) {
      return;
    }
    if (!(s_hi > s_lo)) return;
    // through the hole: rho^2 is convex in s, its maximum over the clipped
    // segment is at an end point
    {
      const double xa = cc.x + s_lo * d.x, ya = cc.y + s_lo * d.y;
      const double xb = cc.x + s_hi * d.x, yb = cc.y + s_hi * d.y;
      const double rin = (R1 - R2) * 0.9999999 - 1e-9;
      if (rin > 0 && xa * xa + ya * ya < rin * rin && xb * xb + yb * yb < rin * rin) return;
    }
    // Crossings of the tube surface by marching the exact distance function
    // f(s) = |(rho - R1, z)| - R2 (|df/ds| <= 1, so a step of |f| cannot
    // jump over the surface), each sign change polished by bracketed Newton.
    // No coefficient arrays, no calls: the state is a dozen registers.
    // Grazing chords make the plain step crawl: it grows by the sine of the grazing angle only
    // (a ray scattered back into the tube at 1e-3 rad needs 2e4 steps; a capped march lost such
    // roots).  A march still running after ODW_TORUS_PLAIN steps therefore uses the curvature
    // bound as well: along the line |f''| <= K = 2 (1/rr + 1/rho) within half of min(rr, rho) of
    // the point (rr = f + R2: distance from the tube's centre circle, rho: from the axis); with
    // F = |f| and G >= (secant slope of |f| over the last step) - K h_last, the surface cannot be
    // reached before (G + sqrt(G^2 + 2 K F)) / K -- sqrt(2 F / K) when running along it.
    const double s_begin = fmax(s_lo, 0.5 * tol - t0);   // roots with t <= tol/2 are rejected anyway
    const double step_min = 1e-7 * R2;
    double sc = s_begin, fc;
    {
      const double px = cc.x + sc * d.x, py = cc.y + sc * d.y, pz = cc.z + sc * d.z;
      const double a = fsqrt(px * px + py * py) - R1;
      fc = fsqrt(a * a + pz * pz) - R2;
    }
    int nfound = 0;
    double h = fmax(fabs(fc), step_min);
    for (int it = 0; it < 16384 && sc < s_hi && nfound < 4; ++it) {
      const double sn = fmin(sc + h, s_hi);
      double fn, rho;
      {
        const double px = cc.x + sn * d.x, py = cc.y + sn * d.y, pz = cc.z + sn * d.z;
        rho = fsqrt(px * px + py * py);
        const double a = rho - R1;
        fn = fsqrt(a * a + pz * pz) - R2;
      }
      h = fmax(fabs(fn), step_min);
      if (it >= ODW_TORUS_PLAIN && (fc > 0) == (fn > 0)) {
        const double hp = sn - sc, rr = fn + R2, room = 0.5 * fmin(rr, rho);
        if (hp <= room) {
          const double K = 2.0 * (frcp(rr) + frcp(rho));
          const double G = (fabs(fn) - fabs(fc)) * frcp(hp) - K * hp;
          const double hs = 0.99 * (G + fsqrt(G * G + 2.0 * K * fabs(fn))) * frcp(K);
          h = fmax(h, fmin(hs, room));            // (a NaN from a pole of the bound drops out)
        }
      }
      if ((fc > 0) != (fn > 0)) {
        double lo = sc, hi = sn, x = sn;
        const bool lo_pos = fc > 0;
        for (int j = 0; j < 48; ++j) {
          const double px = cc.x + x * d.x, py = cc.y + x * d.y, pz = cc.z + x * d.z;
          const double rho = fsqrt(px * px + py * py);
          const double a = rho - R1;
          const double rr = fsqrt(a * a + pz * pz);
          const double f = rr - R2;
          if (f == 0) break;
          if ((f > 0) == lo_pos) lo = x; else hi = x;
          // df/ds = grad f . d,  grad f = ((a/rr) (px,py)/rho, pz/rr)
          const double ir = frcp(rr * rho);
          const double df = (a * (px * d.x + py * d.y)) * ir + pz * d.z * frcp(rr);
          double xn = x - f * frcp(df);
          if (!(xn >= lo && xn <= hi)) xn = 0.5 * (lo + hi);
          // converged when Newton stops moving (it then sits on a bracket end:
          // the bracket test above must be inclusive) or the bracket is empty
          const bool done = xn == x || (hi - lo) <= 4e-16 * (fabs(lo) + fabs(hi));
          x = xn;
          if (done) break;
        }
        const double tr = x + t0;
        if (nfound == 0) c.t0 = tr; else if (nfound == 1) c.t1 = tr; else if (nfound == 2) c.t2 = tr; else c.t3 = tr;
        ++nfound;
      }
      sc = sn;
      fc = fn;
    }
  }
  }
  if (cond_cnt == 0) {
    // untrimmed: only the nearest admissible candidate can win
    double bt = INFINITY;
    int bf = 0;
#if ODW_FLAT_NOBRANCH & 2
#define ODW_PICK(T, F) { const bool w_ = (bool)((int)((T) > tol) & ((int)((T) < bt) | ((int)((T) == bt) & (int)((F) < bf)))); bt = w_ ? (T) : bt; bf = w_ ? (F) : bf; }
#else
#define ODW_PICK(T, F) if ((T) > tol && ((T) < bt || ((T) == bt && (F) < bf))) { bt = (T); bf = (F); }
#endif
    ODW_PICK(c.t0, c.f0)
    ODW_PICK(c.t1, c.f1)
    ODW_PICK(c.t2, c.f2)
    ODW_PICK(c.t3, c.f3)
#undef ODW_PICK
    if constexpr (SPEC::enabled) consider_spec<PARAB, SPEC, PI>(sv, q, bt, bf);
    else consider<PARAB>(sv, q, bt, p, bf, group, 0, 0);
  } else if constexpr (SPEC::enabled) {
    // (a sphere has two candidates; the trimming code exists once per candidate slot)
    // (faces that the boolean left nothing of produce no candidate: their slots are not looked at)
    constexpr int fm = (SPEC::flags(PI) >> ODW_FACEMASK_SHIFT) & 0xff;
    constexpr bool quadric = SPEC::type(PI) == ODW_PRIM_CYLINDER || SPEC::type(PI) == ODW_PRIM_CONE || SPEC::type(PI) == ODW_PRIM_PARABOLOID;
    if constexpr (!ODW_SPEC_LEANCONS || !quadric || (fm & 1) != 0) {
      consider_spec<PARAB, SPEC, PI>(sv, q, c.t0, c.f0);
      consider_spec<PARAB, SPEC, PI>(sv, q, c.t1, c.f1);
    }
    if constexpr (SPEC::type(PI) != ODW_PRIM_SPHERE) {
      if constexpr (!ODW_SPEC_LEANCONS || !quadric || (fm & 2) != 0) consider_spec<PARAB, SPEC, PI>(sv, q, c.t2, c.f2);
      if constexpr (!ODW_SPEC_LEANCONS || !quadric || (fm & 4) != 0) consider_spec<PARAB, SPEC, PI>(sv, q, c.t3, c.f3);
    }
  } else {
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const double t = k == 0 ? c.t0 : (k == 1 ? c.t1 : (k == 2 ? c.t2 : c.t3));
      const int f = k == 0 ? c.f0 : (k == 1 ? c.f1 : (k == 2 ? c.f2 : c.f3));
      consider<PARAB>(sv, q, t, p, f, group, cond_off, cond_cnt);
    }
  }
}

// squared distance of the point w from the segment 0 -> e
__device__ __forceinline__ double seg_dist2(d3 w, d3 e) {
#pragma clang fp contract(off)
  const double s = fmin(fmax(dot(w, e) / dot(e, e), 0.0), 1.0);
  const d3 r = w - e * s;
  return dot(r, r);
}

// One facet of a tessellated face (ODW_PRIM_TRIANGLE, BVH kernels only):
// Moeller-Trumbore in float64; prim_f64 rows = v0, e1, e2, unit facet normal,
// then the barycentric slack per unit of tolerance (a hit within distTol of
// the facet counts, like `dist(p, trimmed face) < distTol`, ray.py:424-426).
__device__ __forceinline__ void intersect_tri(const SceneView& sv, Query& q, int p, int group) {
  cf64 pf = sv.prim_f64 + (size_t)p * 16;
  const d3 e1 = mk(pf[3], pf[4], pf[5]), e2 = mk(pf[6], pf[7], pf[8]);
  const d3 pv = cross(q.dn, e2);
  const double det = dot(e1, pv);
  if (det == 0) return;                              // ray parallel to the facet's plane
  const double inv = frcp(det);
  const d3 tv = q.start - mk(pf[0], pf[1], pf[2]);
  const double u = dot(tv, pv) * inv;
  const d3 qv = cross(tv, e1);
  const double v = dot(q.dn, qv) * inv;
  const double tol = q.tol;
  // per edge: reach of the tolerance in barycentric units; negative = an edge shared with a
  // neighbouring facet of the same face: closed up to rounding
  const double s0 = pf[12], s1 = pf[13], s2 = pf[14];
  if (u < -(s0 < 0 ? 1e-9 : tol * s0) || v < -(s1 < 0 ? 1e-9 : tol * s1) || u + v > 1.0 + (s2 < 0 ? 1e-9 : tol * s2)) return;
  if ((u < 0 && s0 >= 0) || (v < 0 && s1 >= 0) || (u + v > 1.0 && s2 >= 0)) {
    // beyond an edge of the face, inside the parallelogram the bounds above allow (far too long for
    // slivers): the distance to the facet itself decides
#pragma clang fp contract(off)
    const d3 w = e1 * u + e2 * v;                      // hit point - v0, in the facet's plane
    double best = seg_dist2(w, e1);
    best = fmin(best, seg_dist2(w, e2));
    best = fmin(best, seg_dist2(w - e1, e2 - e1));
    if (best > tol * tol) return;
  }
  consider(sv, q, dot(e2, qv) * inv, p, 0, group, 0, 0);
}

// facet or interpolated normal of a triangle at the global point gp
__device__ __forceinline__ d3 tri_normal(cf64 pf, const double* __restrict__ vn, d3 gp) {
  const d3 ng = mk(pf[9], pf[10], pf[11]);
  if (!vn) return ng;
  const d3 e1 = mk(pf[3], pf[4], pf[5]), e2 = mk(pf[6], pf[7], pf[8]);
  const d3 tv = gp - mk(pf[0], pf[1], pf[2]);
  const d3 nn = cross(e1, e2);
  const double inv = frcp(dot(nn, nn));
  const double u = dot(cross(tv, e2), nn) * inv, v = dot(cross(e1, tv), nn) * inv;
  const double w = 1.0 - u - v;
  const d3 n = mk(w * vn[0] + u * vn[3] + v * vn[6], w * vn[1] + u * vn[4] + v * vn[7],
                  w * vn[2] + u * vn[5] + v * vn[8]);
  const double l2 = dot(n, n);
  return l2 > 0 ? n * frsqrt(l2) : ng;
}

// outward normal of face `face` of primitive p at local point lp
template <bool PARAB = true>
__device__ __forceinline__ d3 face_normal(int type, cf64 par, int face, d3 lp) {
  if (type == ODW_PRIM_BOX) {
    const double s = (face & 1) ? 1.0 : -1.0;
    const int a = face >> 1;
    return mk(a == 0 ? s : 0.0, a == 1 ? s : 0.0, a == 2 ? s : 0.0);
  }
  if (type == ODW_PRIM_SPHERE) return lp * frsqrt(dot(lp, lp));
  if (type == ODW_PRIM_CYLINDER || type == ODW_PRIM_CONE || (PARAB && type == ODW_PRIM_PARABOLOID)) {
    if (face == 1) return mk(0, 0, -1);
    if (face == 2) return mk(0, 0, 1);
    double k = 0;
    if (type == ODW_PRIM_CONE) k = (par[1] - par[0]) / par[2];
    // gradient of x^2 + y^2 - (R1 + k z)^2, or of x^2 + y^2 - 4 f z
    const d3 g = mk(lp.x, lp.y, (PARAB && type == ODW_PRIM_PARABOLOID) ? -2.0 * par[0] : -k * (par[0] + k * lp.z));
    return g * frsqrt(dot(g, g));
  }
  const double f = 1.0 - par[0] * frsqrt(lp.x * lp.x + lp.y * lp.y);
  const d3 g = mk(lp.x * f, lp.y * f, lp.z);
  return g * frsqrt(dot(g, g));
}

// slab test against a global AABB (already enlarged by the tolerance slack).
// A NaN from 0*inf drops out of fmin/fmax, i.e. that axis does not cull.
#ifndef ODW_BVH_STACK
#define ODW_BVH_STACK 32
#endif
template <class P>
__device__ __forceinline__ bool ray_box(P bx, d3 oi, d3 inv, double tmax) {
  // oi = origin * inv (component-wise): one fma per plane
  double t0 = fma(bx[0], inv.x, -oi.x), t1 = fma(bx[3], inv.x, -oi.x);
  double lo = fmin(t0, t1), hi = fmax(t0, t1);
  t0 = fma(bx[1], inv.y, -oi.y); t1 = fma(bx[4], inv.y, -oi.y);
  lo = fmax(lo, fmin(t0, t1)); hi = fmin(hi, fmax(t0, t1));
  t0 = fma(bx[2], inv.z, -oi.z); t1 = fma(bx[5], inv.z, -oi.z);
  lo = fmax(lo, fmin(t0, t1)); hi = fmin(hi, fmax(t0, t1));
  return hi >= fmax(lo, 0.0) && lo < tmax;
}

// findNearestIntersection (ray.py:290-452).  Returns prim (<0: none).
// BVH=false: flat loop, wave-uniform primitive index (scalar loads), each
//            primitive culled by its bounding box first -- the analogue of the
//            reference's shell/face BoundBox culls (ray.py:353-398);
// BVH=true : stack traversal, node stack in LDS (one column per thread).
// one primitive of a compiled scene (flat loop unrolled over SPEC)
// boxhit: the outcome of the box test of every primitive so far.  Operands of a Common that must lie inside
// each other (a lens = sphere in cylinder, cylinder in sphere) carry the SAME box -- each one's own box cut by
// the other's -- which shows in the structure: equal sets {primitive} + {primitives it must be inside}
// (SPEC::box_of = the first primitive with that set).  The later ones reuse the first one's outcome (taken
// with a cut at least as wide: conservative).
template <class SPEC, int PI>
__device__ __forceinline__ void spec_prim(const SceneView& sv, Query& q, d3 oi, d3 inv, int skip_solid, int only_solid,
                                          uint64_t mask, bool* boxhit) {
  constexpr int flags = SPEC::flags(PI);
  // relevant groups: a constant without sequential mode (ignored groups' primitives leave no code),
  // the ray's own mask with it
  if constexpr (!SPEC::dead(PI) && (SPEC::seq() || ((SPEC::umask() >> SPEC::group(PI)) & 1) != 0)) {
    // (only_solid: the isolated solid the ray has just entered, ODW_FLAG_ISOLATED -- a scene without such solids
    //  never sets it, and the test folds away)
    if ((!SPEC::seq() || ((mask >> SPEC::group(PI)) & 1)) && (flags >> ODW_SOLID_SHIFT) != skip_solid &&
        (!SPEC::isolated() || only_solid < 0 || (flags >> ODW_SOLID_SHIFT) == only_solid)) {
      bool in_box;
      if constexpr (SPEC::box_of(PI) != PI) {
        in_box = boxhit[SPEC::box_of(PI)];
      } else {
        const double cut = (ODW_SPEC_CUT && ODW_SPEC_KEY) ? q.cut : fmin(q.tmax, q.any.t + 2.0 * q.tol);
        in_box = ray_box(sv.prim_hdr + 8 * PI, oi, inv, cut);
#if ODW_DOUBLE == 1
        in_box = in_box & ray_box(sv.prim_hdr + 8 * PI, mk(opq(oi.x), oi.y, oi.z), inv, cut);
#endif
      }
      boxhit[PI] = in_box;
      if (in_box)
        intersect_prim<SPEC::parab(), SPEC, PI>(sv, q, PI, SPEC::type(PI), SPEC::group(PI), flags, SPEC::cond_word(PI));
#if ODW_DOUBLE == 3 || ODW_DOUBLE == 4 || ODW_DOUBLE == 12
      // (the second pass finds the candidates already known: consider_spec leaves early, before the trimming tests)
      if (in_box && SPEC::type(PI) == (ODW_DOUBLE == 3 ? ODW_PRIM_CYLINDER : (ODW_DOUBLE == 4 ? ODW_PRIM_BOX : ODW_PRIM_SPHERE))) {
        Query q2 = q;
        q2.start.x = opq(q.start.x);
        intersect_prim<SPEC::parab(), SPEC, PI>(sv, q2, PI, SPEC::type(PI), SPEC::group(PI), flags, SPEC::cond_word(PI));
        q.any = q2.any; q.oth = q2.oth;
      }
#endif
    } else if constexpr (SPEC::box_of(PI) == PI) {
      // (skipped for this lane -- not relevant, or the convex solid just left --, but a later primitive may
      //  ask for this box: its own test then)
      boxhit[PI] = SPEC::box_shared(PI) ? ray_box(sv.prim_hdr + 8 * PI, oi, inv, (ODW_SPEC_CUT && ODW_SPEC_KEY) ? q.cut : fmin(q.tmax, q.any.t + 2.0 * q.tol)) : false;
    }
  }
}
template <class T, T... I> struct IndexList {};      // (std::integer_sequence without <utility>: runtime compilation has no libstdc++)
template <class SPEC, int... PI>
__device__ __forceinline__ void spec_prims(const SceneView& sv, Query& q, d3 oi, d3 inv, int skip_solid, int only_solid,
                                           uint64_t mask, IndexList<int, PI...>) {
  bool boxhit[sizeof...(PI)] = {};
  (spec_prim<SPEC, PI>(sv, q, oi, inv, skip_solid, only_solid, mask, boxhit), ...);
}

template <bool BVH, class SPEC = NoSpec>
__device__ __forceinline__ int nearest(const DeviceScene& sc, const SceneView& sv,
                                       const DeviceLimits& lim, d3 start, d3 dn, int medium,
                                       uint64_t mask, double& t_hit, int& face,
                                       int* __restrict__ stack, int skip_solid, int only_solid = -1) {
  Query q;
  q.start = start; q.dn = dn; q.tol = lim.dist_tol; q.tmax = lim.max_ray_length + lim.dist_tol;
  q.medium = medium;
  q.in_medium = __ballot(medium >= 0) != 0ull;
  q.cut = q.tmax;
  q.any.t = INFINITY; q.any.prim = 0x7fffffff; q.any.face = 0x7fffffff;
  q.oth = q.any;
#if ODW_DOUBLE == 9
  d3 inv = mk(frcp1(dn.x), frcp1(dn.y), frcp1(dn.z));
  d3 oi = mk(start.x * inv.x, start.y * inv.y, start.z * inv.z);
  {
    const d3 dn2 = mk(opq(dn.x), opq(dn.y), opq(dn.z));
    const d3 inv2 = mk(frcp1(dn2.x), frcp1(dn2.y), frcp1(dn2.z));
    const d3 oi2 = mk(start.x * inv2.x, start.y * inv2.y, start.z * inv2.z);
    inv = inv2.x == inv.x ? inv : inv2;
    oi = oi2.y == oi.y ? oi : oi2;
  }
#else
  const d3 inv = mk(frcp1(dn.x), frcp1(dn.y), frcp1(dn.z));
  const d3 oi = mk(start.x * inv.x, start.y * inv.y, start.z * inv.z);
#endif
  if constexpr (SPEC::enabled) {
    spec_prims<SPEC>(sv, q, oi, inv, skip_solid, only_solid, mask, __make_integer_seq<IndexList, int, SPEC::N>{});
  } else if (!BVH) {
    // without sequential mode the set of relevant groups is the same for
    // every ray: the test is scalar and skips a primitive for the whole wave
    const bool per_lane_mask = sc.seq_enabled != 0;
    const uint64_t umask = sc.all_mask & ~sc.ignore_mask;
    for (int p = 0; p < sc.n_prims; ++p) {
      cf64 hdr = sv.prim_hdr + 8 * p;                 // one 64-byte scalar load
      ci32 hi = (ci32)(hdr + 6);
      const int type = hi[0], g = hi[1], flags = hi[2], cond_word = hi[3];
      if (per_lane_mask) {
        if (!((mask >> g) & 1)) continue;
      } else {
        if (!((umask >> g) & 1)) continue;
      }
      // a ray that has just left a convex solid cannot meet it again; one that has just entered an isolated
      // solid meets that solid first (ODW_FLAG_ISOLATED)
      if ((flags >> ODW_SOLID_SHIFT) == skip_solid) continue;
      if (only_solid >= 0 && (flags >> ODW_SOLID_SHIFT) != only_solid) continue;
      // candidates farther than the nearest hit + 2*distTol can never be
      // selected (ray.py:432,440): shrink the search like the reference does
      const double cut = fmin(q.tmax, q.any.t + 2.0 * q.tol);
      if (!ray_box(hdr, oi, inv, cut)) continue;
      intersect_prim<false>(sv, q, p, type, g, flags, cond_word);
    }
  } else {
    // BVH traversal in float32 (culling only: leaves are intersected in
    // float64).  One 64-byte node fetch carries the boxes of both children.
    // "while-while": every lane walks inner nodes (cheap, same code for all
    // lanes) until it stands on a leaf or is done; only then are the leaf
    // primitives intersected (expensive), so that part runs with as many
    // lanes as possible.  A lane that reaches its first leaf before the others
    // parks it and walks on while some lane still looks for its first
    // (hugeArray +9 %, meshes unchanged).  Stack entries: >= 0 inner node, < -1 leaf
    // (~(first | count << 24) - 1); node stack in LDS, one column per thread.
    typedef const float ODW_CONST* cf32;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    typedef int vi4 __attribute__((ext_vector_type(4)));
    cf32 nodes = (cf32)(uintptr_t)sc.bvh_nodes;
    ci32 bvh_prims = as_const(sc.bvh_prims);
    const float ofx = (float)start.x, ofy = (float)start.y, ofz = (float)start.z;
    const float ivx = (float)inv.x, ivy = (float)inv.y, ivz = (float)inv.z;
    int sp = 0;
    int cur = 0;
    int pend = -1;
    for (;;) {
      // (a lane with a parked leaf walks on only while some lane still looks for its first)
      for (;;) {
        // (the vote is taken by every lane still in this loop, before any of them leaves)
        const unsigned long long looking = __ballot(cur >= 0 && pend == -1);
        if (!(cur >= 0 && (pend == -1 || looking != 0ull))) break;
        cf32 nd = nodes + (size_t)cur * 16;
        const vf4 a0 = *reinterpret_cast<const vf4 ODW_CONST*>(nd);
        const vf4 a1 = *reinterpret_cast<const vf4 ODW_CONST*>(nd + 4);
        const vf4 a2 = *reinterpret_cast<const vf4 ODW_CONST*>(nd + 8);
        const vi4 lk = *reinterpret_cast<const vi4 ODW_CONST*>(nd + 12);
        // lo0 = a0.xyz, hi0 = (a0.w, a1.x, a1.y), lo1 = (a1.z, a1.w, a2.x), hi1 = a2.yzw
        const float cutf = (float)fmin(q.tmax, q.any.t + 2.0 * q.tol) * 1.00001f + 1e-3f;
        float tn0, tf0, tn1, tf1;
        {
          const float x0 = (a0.x - ofx) * ivx, x1 = (a0.w - ofx) * ivx;
          const float y0 = (a0.y - ofy) * ivy, y1 = (a1.x - ofy) * ivy;
          const float z0 = (a0.z - ofz) * ivz, z1 = (a1.y - ofz) * ivz;
          tn0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
          tf0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        }
        {
          const float x0 = (a1.z - ofx) * ivx, x1 = (a2.y - ofx) * ivx;
          const float y0 = (a1.w - ofy) * ivy, y1 = (a2.z - ofy) * ivy;
          const float z0 = (a2.x - ofz) * ivz, z1 = (a2.w - ofz) * ivz;
          tn1 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
          tf1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        }
        // conservative acceptance: relative 1e-5 + absolute 1e-3 mm on t
        const bool h0 = tf0 * 1.00001f + 1e-3f >= fmaxf(tn0, 0.f) * 0.99999f - 1e-3f && tn0 * 0.99999f - 1e-3f < cutf;
        const bool h1 = tf1 * 1.00001f + 1e-3f >= fmaxf(tn1, 0.f) * 0.99999f - 1e-3f && tn1 * 0.99999f - 1e-3f < cutf;
        const int r0 = lk.z > 0 ? -2 - (lk.x | (lk.z << 24)) : lk.x;
        const int r1 = lk.w > 0 ? -2 - (lk.y | (lk.w << 24)) : lk.y;
        if (h0 && h1) {
          const bool first0 = tn0 <= tn1;
          stack[sp * 256] = first0 ? r1 : r0;
          ++sp;
          cur = first0 ? r0 : r1;
        } else if (h0 || h1) {
          cur = h0 ? r0 : r1;
        } else if (sp > 0) {
          --sp;
          cur = stack[sp * 256];
        } else {
          cur = -1;
        }
        // a lane that reaches its first leaf parks it and keeps walking while
        // the rest of the wave still walks (it would idle otherwise)
        if (cur < -1 && pend == -1) {
          pend = cur;
          if (sp > 0) { --sp; cur = stack[sp * 256]; } else cur = -1;
        }
      }
      int lf;
      if (pend != -1) { lf = pend; pend = -1; }
      else if (cur < -1) {
        lf = cur;
        if (sp > 0) { --sp; cur = stack[sp * 256]; } else cur = -1;
      } else break;
      const int code = -2 - lf;
      const int leaf_first = code & 0xffffff, leaf_count = code >> 24;
      for (int i = 0; i < leaf_count; ++i) {
        const int p = bvh_prims[leaf_first + i];
        ci32 pi = sv.prim_i32 + 4 * p;
        const int g = pi[1];
        if (((mask >> g) & 1) && (pi[2] >> ODW_SOLID_SHIFT) != skip_solid) {
          if (pi[0] == ODW_PRIM_TRIANGLE) intersect_tri(sv, q, p, g);
          else intersect_prim<true>(sv, q, p, pi[0], g, pi[2], pi[3]);
        }
      }
    }
  }
  if constexpr (SPEC::enabled && ODW_SPEC_KEY) {
    // (consider_spec keeps primitive << 8 | face in the `face` member)
    if (q.any.face == 0x7fffffff) return -1;
    const bool use_oth_ = q.oth.face != 0x7fffffff && q.oth.t < q.any.t + 2.0 * q.tol;
    t_hit = use_oth_ ? q.oth.t : q.any.t;
    const int key = use_oth_ ? q.oth.face : q.any.face;
    face = key & 0xff;
    return key >> 8;
  }
  if (q.any.prim == 0x7fffffff) return -1;
  // hits within 2*distTol of the nearest: prefer one whose group differs
  // from the current medium (ray.py:438-452)
  // (selected member by member: a reference picked with ?: would pin both
  // records in scratch memory)
  const bool use_oth = q.oth.prim != 0x7fffffff && q.oth.t < q.any.t + 2.0 * q.tol;
  t_hit = use_oth ? q.oth.t : q.any.t;
  face = use_oth ? q.oth.face : q.any.face;
  return use_oth ? q.oth.prim : q.any.prim;
}

// ------------------------------------------ mirror / Snell / grating
__device__ __forceinline__ d3 mirror(d3 r, d3 n) { return r - n * (2.0 * dot(r, n)); }

__device__ __forceinline__ d3 snells_law(d3 r, double n1, double n2, d3 n, bool& tir) {
  // ray.py:488-495 writes root = 1 - mu^2 |n x r|^2 and the transmitted direction mu n x ((-n) x r) + n sqrt(root).
  // For the unit vectors this is called with, |n x r|^2 = 1 - (n.r)^2 and n x ((-n) x r) = r - n (n.r): three cross
  // products less, the same numbers up to rounding (the oracle keeps the reference's form; 1e-16, inside every
  // tolerance of the parity tests) -- 25 of the ~75 float64 instructions of a refraction
  const double d = dot(n, r);
  const double mu = n1 * frcp(n2);
  const double root = 1.0 - mu * mu * (1.0 - d * d);
  if (root < 0) { tir = true; return r - n * (2.0 * d); }
  tir = false;
  const d3 perp = r - n * d;
  return perp * mu + n * fsqrt(root);
}

__device__ __forceinline__ d3 line_grating(d3 ray, double n1, double n2, d3 normal, double wavelength_nm,
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

// ------------------------------------------- stochastic surfaces (N3)
// FreeCAD Rotation(axis, angle) * v; a zero axis is the identity
__device__ __forceinline__ d3 rotate(d3 axis, double angle, d3 v) {
  const double l2 = dot(axis, axis);
  if (l2 == 0) return v;
  const d3 k = axis * (1.0 / sqrt(l2));
  double s, c;
  sincos_bounded(angle, s, c);
  return v * c + cross(k, v) * s + k * (dot(k, v) * (1.0 - c));
}
__device__ __forceinline__ double acos_clamped(double x) { return acos(fmax(-1.0, fmin(1.0, x))); }

__device__ __forceinline__ void surface_draw(const DeviceSurfaceSampler* sp, double theta_in, double theta_refl,
                                             uint64_t ray, uint64_t seed, uint32_t ordinal, uint32_t stream,
                                             double& theta, double& phi) {
  const DeviceSurfaceSampler S = *sp;
  int k = 0;
  if (S.axis != ODW_SURF_AXIS_NONE) {
    // The reference compiles its tables at the hit's own value of the constant; here the two family
    // members around it are mixed: member k0 + 1 with probability = the fractional position between
    // the knots (a linear interpolation of the two distributions in the constant), decided by a
    // uniform of its own.  At a knot the member is the reference's table bit for bit.
    const double c = S.axis == ODW_SURF_AXIS_THETA_IN ? theta_in : theta_refl;
    const double kf = fmin(fmax((c - S.lo) * S.inv_step, 0.0), (double)(S.n_family - 1));
    const int k0 = (int)floor(kf);
    const double frac = kf - (double)k0;
    uint32_t m0 = (uint32_t)ray, m1 = (uint32_t)(ray >> 32), m2 = ordinal, m3 = stream + 16u;
    philox4x32_10(m0, m1, m2, m3, (uint32_t)seed, (uint32_t)(seed >> 32));
    k = k0 + (u53(m0, m1) < frac ? 1 : 0);
  }
  uint32_t c0 = (uint32_t)ray, c1 = (uint32_t)(ray >> 32), c2 = ordinal, c3 = stream;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
  double u_phi = u53(c0, c1);
  const double u_t = u53(c2, c3);
  if (S.n_atoms) {
    // discrete events (DiracDelta terms): atom j owns [sum of the masses before it, + its own) of u_phi; what is
    // left of the unit interval goes on as the tables' u_phi
    double acc = 0.0;
    for (int j = 0; j < S.n_atoms; ++j) {
      const double pj = S.atom_mass[(size_t)k * (size_t)S.n_atoms + j];
      if (u_phi < acc + pj) {
        theta = S.atom_theta[j][0] + S.atom_theta[j][1] * theta_in + S.atom_theta[j][2] * theta_refl;
        phi = S.atom_phi[j][0] != 0.0 ? S.atom_phi[j][1] + (u_phi - acc) / pj * (S.atom_phi[j][2] - S.atom_phi[j][1]) : S.atom_phi[j][1];
        return;
      }
      acc += pj;
    }
    u_phi = fmin((u_phi - acc) / (1.0 - acc), 0.99999999999999989);
  }
  TableView tv;
  tv.phi_tab = S.phi_tab + (size_t)k * (size_t)S.n_phi_knots * 2;
  tv.t_tab = S.t_tab + (size_t)k * (size_t)S.n_t_rows * (size_t)S.n_t_knots * 2;
  tv.t_guide = S.t_guide + (size_t)k * (size_t)S.n_t_rows * (size_t)(S.n_guide + 1);
  tv.n_phi_knots = S.n_phi_knots; tv.n_t_knots = S.n_t_knots; tv.n_t_rows = S.n_t_rows; tv.n_guide = S.n_guide;
  tv.phi_guide = nullptr; tv.n_phi_guide = 0;
  sample_tables(tv, u_phi, u_t, theta, phi);
}

// the sampler of a (group, kind) chain that serves a hit with n1 / n2 = mu (mu <= 0: total reflection): the one
// whose own mu is nearest in log mu; samplers with mu = 0 serve every hit
__device__ __forceinline__ int pick_sampler(const DeviceSurfaceSampler* samplers, int first, double mu) {
  int best = first;
  double dist = INFINITY;
  for (int j = first; j >= 0; j = samplers[j].next) {
    const double m = samplers[j].mu;
    if (m == 0.0) return j;
    const double d = (m < 0 || mu <= 0) ? ((m < 0) == (mu <= 0) ? 0.0 : INFINITY) : fabs(log(m / mu));
    if (d < dist) { dist = d; best = j; }
  }
  return best;
}

// OpticalGroupProxy.applyStochasticRayCorrections (optical_group.py:279-323)
// mu: n1 / n2 of a lens hit, <= 0 for total reflection; anything for mirrors
__device__ __noinline__ d3 scatter(const DeviceSurfaceSampler* samplers, int s_primary, int s_modify,
                                   uint64_t ray, uint64_t seed, uint32_t ordinal, d3 din, d3 ideal, d3 n, double mu) {
  if (s_primary < 0 && s_modify < 0) return ideal;
  if (s_primary >= 0 && samplers[s_primary].next >= 0) s_primary = pick_sampler(samplers, s_primary, mu);
  const double nl = sqrt(dot(n, n));
  const double theta_in = acos_clamped(dot(din, n) / nl);
  const double theta_refl = acos_clamped(dot(ideal, n) / (sqrt(dot(ideal, ideal)) * nl));
  d3 out = ideal;
  double theta, phi;
  if (s_primary >= 0) {
    surface_draw(samplers + s_primary, theta_in, theta_refl, ray, seed, ordinal, 1u + ODW_SURF_PRIMARY, theta, phi);
    out = rotate(n, phi, rotate(cross(n, din), theta, n));
  }
  if (s_modify >= 0) {
    surface_draw(samplers + s_modify, theta_in, theta_refl, ray, seed, ordinal, 1u + ODW_SURF_MODIFY, theta, phi);
    out = rotate(out, phi, rotate(cross(out, din), theta, out));
  }
  return out * (1.0 / sqrt(dot(out, out)));
}

// ------------------------------------------------------- recording (K5)
// Active lanes append one 64-B row each: lanes take consecutive slots by
// popcount prefix of the ballot, the slots come from the wave's reserved block
// (big lists) or from one atomic per append (small lists, exact fill).
// block change of a wave's hit-list reservation (rare: out of line): the leader takes a new block.  The lanes that
// still fit fill the rest of the old block, the others start the new one (round 4; before, the < 64 slots the old
// block had left were tagged unused: 6 % of a big list's slots, written as tags and read again by every pass over
// the list)
__device__ __noinline__ void next_hit_block(unsigned long long* hit_count, uint32_t block, volatile uint32_t* hit_state, bool leader) {
  if (leader) {
    const unsigned long long base = atomicAdd(hit_count, (unsigned long long)block);
    hit_state[0] = (uint32_t)base;
    hit_state[1] = (uint32_t)(base >> 32);
  }
}

// cnt: the thread's (CS = 256: a column of the block's table, plain adds) or the wave's (CS = 1,
// CA: ds_add) event counters
// Detector histogram: the counts are u64 in HBM, and a global atomic executes at the memory side
// (every add one uncached request; adds to the same few lines queue there).  A focused beam puts most
// of its hits into a few thousand bins -- 1e8 such adds cap a launch at ~6.5e9 rays/s (measured) --, so
// the flat kernels keep a WINDOW of ODW_HIST_WIN x ODW_HIST_WIN bins of the histogram per block in LDS
// (u32 counts, ds_add): the first recorded hit of the block centres it, hits inside it cost an LDS
// add, hits outside still go to HBM, and every block adds its window to the histogram once, when it
// ends.  Integer sums: the histogram is the same whichever way a hit was counted.
// win: [0] 0 = no window yet, ~0 = being opened, else first bin in x + 1; [1] first bin in y;
//      [4 ...] counts (x major)
#ifndef ODW_HIST_WIN
#define ODW_HIST_WIN 88
#endif
// the block's first recorded hit opens the window around its bin (once per block and launch: out of line)
// (around the MEAN bin of the hits that arrive with the first one -- the lanes of the wave that record together, and
//  whatever other waves add before the opener reads: win[2] = sum of x | count << 20, win[3] the same for y, each
//  word consistent in itself.  A window around one hit is off by the beam's own width)
__device__ __forceinline__ void hist_window_open(uint32_t* win, int ix, int iy, int nx, int ny) {
  atomicAdd(win + 2, (uint32_t)min(ix, 4095) | (1u << 20));     // (256 lanes x 4095 < 2^20)
  atomicAdd(win + 3, (uint32_t)min(iy, 4095) | (1u << 20));
  if (atomicCAS(win, 0u, ~0u) == 0u) {
    volatile uint32_t* vw = win;
    const uint32_t sx = vw[2], sy = vw[3];
    const int mx = (sx >> 20) ? (int)((sx & 0xfffffu) / (sx >> 20)) : ix, my = (sy >> 20) ? (int)((sy & 0xfffffu) / (sy >> 20)) : iy;
    vw[1] = (uint32_t)max(0, min(my - ODW_HIST_WIN / 2, ny - ODW_HIST_WIN));
    vw[0] = (uint32_t)max(0, min(mx - ODW_HIST_WIN / 2, nx - ODW_HIST_WIN)) + 1u;
  }
}
// true: the hit was counted in the window
__device__ __forceinline__ bool hist_window_add(uint32_t* win, int ix, int iy, int nx, int ny) {
  volatile uint32_t* vw = win;
  uint32_t x1 = vw[0];
  if (x1 == 0u) {
    hist_window_open(win, ix, iy, nx, ny);
    x1 = vw[0];
  }
  if (x1 == 0u || x1 == ~0u) return false;     // another wave is opening it: this hit goes to HBM
  const uint32_t dx = (uint32_t)ix - (x1 - 1u), dy = (uint32_t)iy - vw[1];
  if (dx >= (uint32_t)ODW_HIST_WIN || dy >= (uint32_t)ODW_HIST_WIN) return false;
  atomicAdd(win + 4 + dx * ODW_HIST_WIN + dy, 1u);
  return true;
}
// PT: TraceParams, or TraceParams in the constant address space (the kernel's argument segment)
template <bool BLOCKS, int CS = 256, bool CA = false, class PT>
__device__ __forceinline__ void record_hit(const PT& P, uint64_t ray, int group, d3 p, d3 d,
                                           double power, bool entering, uint32_t* cnt,
                                           volatile uint32_t* hit_state, uint32_t* win = nullptr) {
  if (P.flags & ODW_TRACE_RECORD_HITS) {
    const uint64_t active = __ballot(1);
    const int lane = __lane_id();
    const int leader = __ffsll((unsigned long long)active) - 1;
    const uint32_t n_act = __popcll(active);
    const uint32_t rank = __popcll(active & ((1ull << lane) - 1ull));
    uint64_t slot;
    // a batch launch (flat kernels): the wave's scene (hit_state[3]) has its own segment of the list and its own counters
    odw_hit* hits = P.out.hits;
    unsigned long long* hit_count = P.out.hit_count;
    uint32_t* row_of = nullptr;
    double* pts = nullptr;
    if (BLOCKS && P.batch.n_scenes) {
      const uint32_t scene = __builtin_amdgcn_readfirstlane(hit_state[3]);
      hits += (size_t)scene * P.out.hit_capacity;
      hit_count += 4 * scene;
      if (P.out.row_of) row_of = P.out.row_of + (size_t)scene * P.out.row_stride;
      if (P.out.pts) pts = P.out.pts + (size_t)scene * P.out.hit_capacity * 3;
      const uint64_t leaving = __ballot(!entering);
      if (leaving && lane == leader) atomicAdd(hit_count + 2, (unsigned long long)__popcll(leaving));
    }
    if (BLOCKS && P.out.hit_block) {
      // One atomic per wave and append serialises 1.5e6 appends of a launch on a single counter at
      // the memory side (1.2 - 3 ms of a 20 ms launch, measured).  A wave therefore takes hit_block
      // slots at a time and hands them out itself; slots it cannot use (< 64 at a block change, the
      // rest of its last block at the end) are marked ODW_TAG_UNUSED and dropped by odw_fetch_hits.
      // the wave's block lives in LDS (base lo, base hi, used): only the recording lanes are active
      // here, a register copy would go stale in the others
      const uint32_t hit_used = hit_state[2];
      const uint32_t left = P.out.hit_block - hit_used;          // (0: no block reserved yet, or the block is full)
      slot = (((uint64_t)hit_state[1] << 32) | hit_state[0]) + hit_used + rank;
      if (n_act > left) {
        next_hit_block(hit_count, P.out.hit_block, hit_state, lane == leader);
        if (rank >= left) slot = (((uint64_t)hit_state[1] << 32) | hit_state[0]) + (rank - left);
        if (lane == leader) hit_state[2] = n_act - left;
      } else if (lane == leader) {
        hit_state[2] = hit_used + n_act;
      }
    } else {
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(hit_count, (unsigned long long)n_act);
      const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
      const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
      slot = (((uint64_t)bhi << 32) | blo) + rank;
    }
    if (slot < P.out.hit_capacity) {
      double2* row = reinterpret_cast<double2*>(hits + slot);
      const uint64_t tag = (ray & 0xFFFFFFFFFFFFull) | ((uint64_t)group << 48) | ((uint64_t)entering << 63);
      // (streamed once, read back much later if at all: non-temporal stores, +0.6 % on C3)
      typedef double vd2 __attribute__((ext_vector_type(2)));
      vd2* rw = reinterpret_cast<vd2*>(row);
      __builtin_nontemporal_store((vd2){p.x, p.y}, rw);
      __builtin_nontemporal_store((vd2){p.z, d.x}, rw + 1);
      __builtin_nontemporal_store((vd2){d.y, d.z}, rw + 2);
      __builtin_nontemporal_store((vd2){power, __longlong_as_double((long long)tag)}, rw + 3);
      if (BLOCKS && row_of) row_of[ray - P.first_ray] = (uint32_t)slot;
      if (BLOCKS && pts) {
        // (component-major: a wave's 64 rows write 512 contiguous bytes per component)
        double* q = pts + slot;
        __builtin_nontemporal_store(p.x, q);
        __builtin_nontemporal_store(p.y, q + P.out.hit_capacity);
        __builtin_nontemporal_store(p.z, q + 2 * P.out.hit_capacity);
      }
    } else if (CA) {
      atomicAdd(&cnt[ODW_CNT_HITS_DROPPED * CS], 1u);
    } else {
      cnt[ODW_CNT_HITS_DROPPED * CS] += 1u;
    }
  }
  if ((P.flags & ODW_TRACE_HISTOGRAM) && P.det_enabled) {
#pragma clang fp contract(off)
    cdetector det = as_const(opaque(P.det));
    if (det->group < 0 || det->group == group) {
      const d3 r = p - mk(det->origin[0], det->origin[1], det->origin[2]);
      const double x = dot(r, mk(det->ex[0], det->ex[1], det->ex[2]));
      const double y = dot(r, mk(det->ey[0], det->ey[1], det->ey[2]));
      const double fx = floor((x - det->x_lo) * det->x_scale);
      const double fy = floor((y - det->y_lo) * det->y_scale);
      if (fx >= 0 && fx < det->nx_f && fy >= 0 && fy < det->ny_f) {
        const int ix = (int)fx, iy = (int)fy;
        if (!(win && hist_window_add(win, ix, iy, det->nx, det->ny)))
          atomicAdd(P.out.hist + ((size_t)ix * (size_t)det->ny + (size_t)iy), 1ull);
      } else if (CA)
        atomicAdd(&cnt[ODW_CNT_HIST_OVERFLOW * CS], 1u);
      else
        cnt[ODW_CNT_HIST_OVERFLOW * CS] += 1u;
    }
  }
}

// K5 of the flat kernels, out of line: a ray records once or twice in its life, and inlined into the
// recording branches this code decides the register allocation of the whole ray loop (30 more spilled
// VGPRs with the histogram window, measured).  kargs: the kernel's argument segment (TraceParams is
// the kernels' only argument; the pointer is taken in the kernel, a callee cannot ask for it).
typedef const TraceParams ODW_CONST* ckargs;
__device__ __noinline__ void record_hit_flat(ckargs kargs, uint64_t ray, int group, d3 p, d3 d, double power,
                                             bool entering, uint32_t* cnt, uint32_t* hit_state, uint32_t* win) {
  // (arguments arrive in vector registers: the pointer is wave-uniform, say so -- scalar loads again)
  const uint64_t a = (uint64_t)(uintptr_t)kargs;
  // (the builtin returns int: without the casts the low word is sign-extended over the high one)
  const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
  record_hit<true>(*(ckargs)(uintptr_t)u, ray, group, p, d, power, entering, cnt, hit_state, win);
}

// RecordRays (generic_source.py:78-118): one row per segment Ray.traceRay yields.  Only a
// handful of rays are recorded this way (the reference draws and pickles them one by one),
// so: one atomic per append, and kernels of their own (SEG) that the bulk launches never run.
__device__ __noinline__ void record_segment(odw_segment* segs, uint64_t capacity, unsigned long long* seg_count,
                                            uint64_t ray, int ordinal, int medium, d3 p1, d3 p2, double power) {
  const uint64_t slot = atomicAdd(seg_count, 1ull);
  if (slot < capacity) {
    double2* row = reinterpret_cast<double2*>(segs + slot);
    const uint64_t tag = (ray & 0xFFFFFFFFFFull) | ((uint64_t)(ordinal & 0xFFF) << 40) |
                         ((uint64_t)((medium + 1) & 0xFFF) << 52);
    row[0] = make_double2(p1.x, p1.y);
    row[1] = make_double2(p1.z, p2.x);
    row[2] = make_double2(p2.y, p2.z);
    row[3] = make_double2(power, __longlong_as_double((long long)tag));
  }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

struct RayInit { d3 point, dir; double power; };
__device__ __noinline__ RayInit generate_ray(const DeviceSource* sp, uint64_t ray, uint64_t seed) {
  RayInit r;
  double up, ut, t, phi;
  csource src = as_const(sp);
  ray_uniforms(ray, seed, up, ut);
  sample_source(src, up, ut, t, phi);
  make_ray(src, t, phi, r.point, r.dir);
  r.power = src->power;
  return r;
}

// K3 for one hit, the part that depends on the optical group: recording, mirror / Snell / absorb /
// vacuum / grating, medium and sequence state (ray.py:91-281).  The generic kernels pass the group's
// words as read from the tables; a compiled scene passes constants and the branches fold.
// n: surface normal along the travel direction; cnt: the thread's column of event counters.
template <bool BVH, bool STOCH, bool LEAN>
__device__ __forceinline__ void interact(const TraceParams& P, cf64 group_f64, ci32 group_i32, cf64 group_gdir, int g,
                                         int gtype, bool record, d3 n, bool entering, uint64_t ray, int nint,
                                         uint32_t* cnt, uint32_t* hit_state, uint32_t* win, d3 point, d3& dir, double& power,
                                         int& medium, int& seq, bool& alive) {
  if (record) {
    cnt[ODW_CNT_RECORDED_HITS * 256] += 1u;
    // (block reservations and the histogram window only in the flat kernels: in the BVH kernels
    // their state costs more registers / LDS than the atomics cost time)
    if (BVH) record_hit<false>(P, ray, g, point, dir, power, entering, cnt, hit_state);
    else record_hit_flat((ckargs)__builtin_amdgcn_kernarg_segment_ptr(), ray, g, point, dir, power, entering, cnt,
                         hit_state, win);   // (interact is inlined into the kernel: the pointer is the kernel's)
  }
  if (gtype == ODW_OPT_MIRROR) {
#if ODW_DOUBLE == 7
    d3 ideal = mirror(dir, n);
    { const d3 again = mirror(mk(opq(dir.x), dir.y, dir.z), n); ideal = again.x == ideal.x ? ideal : again; }
#else
    const d3 ideal = mirror(dir, n);
#endif
    if (STOCH) dir = scatter(P.samplers, P.group_sampler[2 * g], P.group_sampler[2 * g + 1], ray, P.seed, (uint32_t)nint,
                             dir, ideal, n, 1.0);
    else dir = ideal;
    power *= group_f64[4 * g + 1];
    ++seq;
  } else if (gtype == ODW_OPT_LENS) {
    const double n1 = (medium >= 0) ? group_f64[4 * medium] : 1.0;
    double n2 = 1.0;
    if (entering) { medium = g; n2 = group_f64[4 * g]; }
    bool tir;
#if ODW_DOUBLE == 7
    d3 ideal = snells_law(dir, n1, n2, n, tir);
    { bool tir2; const d3 again = snells_law(mk(opq(dir.x), dir.y, dir.z), n1, n2, n, tir2); ideal = again.x == ideal.x ? ideal : again; tir = tir & tir2; }
#else
    const d3 ideal = snells_law(dir, n1, n2, n, tir);
#endif
    if (STOCH) dir = scatter(P.samplers, P.group_sampler[2 * g], P.group_sampler[2 * g + 1], ray, P.seed, (uint32_t)nint,
                             dir, ideal, n, tir ? -1.0 : n1 / n2);
    else dir = ideal;
    if (!entering && !tir && medium == g) { medium = -1; ++seq; }
  } else if (gtype == ODW_OPT_ABSORBER) {
    power = 0;
    ++seq;
  } else if (LEAN || gtype == ODW_OPT_VACUUM) {
    ++seq;
  } else {  // grating (ray.py:216-268)
    const d3 gd = mk(group_gdir[3 * g], group_gdir[3 * g + 1], group_gdir[3 * g + 2]);
    const double lpm = group_f64[4 * g + 3];
    const int order = group_i32[4 * g + 3];
    if (group_i32[4 * g + 2] == 0) {
      if (entering) {
        const double nn = (medium >= 0) ? group_f64[4 * medium] : 1.0;
        dir = line_grating(dir, nn, nn, n, P.wavelength, order, lpm, gd, false);
        ++seq;
      }
    } else if (entering) {
      if (medium >= 0) {
        // the reference raises ValueError here (ray.py:234-237): counted, the host raises
        atomicAdd(P.out.counters + ODW_CNT_GRATING_IN_MEDIUM, 1ull);
        cnt[ODW_CNT_DIED * 256] += 1u;
        alive = false;
      }
      medium = g;
      dir = line_grating(dir, 1.0, group_f64[4 * g], n, P.wavelength, order, lpm, gd, true);
    } else {
      const double n1 = (medium >= 0) ? group_f64[4 * medium] : 1.0;
      bool tir;
      dir = snells_law(dir, n1, 1.0, n, tir);
      if (!tir) { medium = -1; ++seq; }
    }
  }
}

// the hit is on primitive PI of a compiled scene: its type, frame pattern, flags, group and the group's
// optical type / recording switch are constants
template <bool STOCH, bool LEAN, class SPEC, int PI>
__device__ __forceinline__ void spec_hit(const TraceParams& P, const SceneView& sv, cf64 group_f64, ci32 group_i32,
                                         cf64 group_gdir, int face, uint64_t ray, int nint, uint32_t* cnt,
                                         uint32_t* hit_state, uint32_t* win, d3 point, d3& dir, double& power, int& medium, int& seq,
                                         int& skip, int& only, bool& alive) {
  constexpr int flags = SPEC::flags(PI), g = SPEC::group(PI);
  cf64 pf = sv.prim_f64 + (size_t)PI * 16;
  d3 n = face_normal<SPEC::parab()>(SPEC::type(PI), pf + 12, face, xf_point_nz<SPEC::xf(PI)>(pf, point));
  if constexpr ((flags & ODW_FLAG_FLIP_NORMAL) != 0) n = n * -1.0;
  n = xf_vec_t_nz<SPEC::xf(PI)>(pf, n);
#if ODW_DOUBLE == 6
  {
    d3 n2 = face_normal<SPEC::parab()>(SPEC::type(PI), pf + 12, face, xf_point_nz<SPEC::xf(PI)>(pf, mk(opq(point.x), point.y, point.z)));
    if constexpr ((flags & ODW_FLAG_FLIP_NORMAL) != 0) n2 = n2 * -1.0;
    n2 = xf_vec_t_nz<SPEC::xf(PI)>(pf, n2);
    n = n2.x == n.x ? n : n2;
  }
#endif
  const bool entering = dot(dir, n) < 0;
  if (entering) n = n * -1.0;
  interact<false, STOCH, LEAN>(P, group_f64, group_i32, group_gdir, g, SPEC::gtype(g), SPEC::record(g), n, entering, ray,
                               nint, cnt, hit_state, win, point, dir, power, medium, seq, alive);
  // n points along the incoming travel direction: out of the solid when leaving it, into it when entering
  const double out = entering ? -dot(dir, n) : dot(dir, n);
  if constexpr ((flags & ODW_FLAG_CONVEX) != 0) skip = out > 0 ? (flags >> ODW_SOLID_SHIFT) : -1;
  else skip = -1;
  if constexpr ((flags & ODW_FLAG_ISOLATED) != 0) only = out < 0 ? (flags >> ODW_SOLID_SHIFT) : -1;
  else only = -1;
}
template <bool STOCH, bool LEAN, class SPEC, int... PI>
__device__ __forceinline__ void spec_hits(const TraceParams& P, const SceneView& sv, cf64 group_f64, ci32 group_i32,
                                          cf64 group_gdir, int prim, int face, uint64_t ray, int nint, uint32_t* cnt,
                                          uint32_t* hit_state, uint32_t* win, d3 point, d3& dir, double& power, int& medium, int& seq,
                                          int& skip, int& only, bool& alive, IndexList<int, PI...>) {
  (void)((prim == PI ? (spec_hit<STOCH, LEAN, SPEC, PI>(P, sv, group_f64, group_i32, group_gdir, face, ray, nint, cnt, hit_state, win,
                                                  point, dir, power, medium, seq, skip, only, alive), true)
                     : false) || ...);
}

// LDS of the hit-list block reservations: exists only where it is used
// (+ the block's histogram window, see record_hit: header at word 16, counts from word 20)
// a wave gives up its block of hit-list slots (end of the kernel; BATCH: the wave moves on to another scene): the
// slots it never filled are tagged unused and counted, the wave's state says "no block"
__device__ __forceinline__ void close_hit_block(const TraceParams& P, volatile uint32_t* hit_state, uint32_t next_scene) {
  if (!P.out.hit_block) {
    if (P.batch.n_scenes && __lane_id() == 0) hit_state[3] = next_scene;   // (lists without block reservations: only the scene word matters)
    return;
  }
  odw_hit* hits = P.out.hits;
  unsigned long long* hit_count = P.out.hit_count;
  if (P.batch.n_scenes) {
    const uint32_t cur = __builtin_amdgcn_readfirstlane(hit_state[3]);
    hits += (size_t)cur * P.out.hit_capacity;
    hit_count += 4 * cur;
  }
  const uint32_t hit_used = hit_state[2];
  const uint64_t hit_base = ((uint64_t)hit_state[1] << 32) | hit_state[0];
  if (hit_used < P.out.hit_block) {
    const uint32_t left = P.out.hit_block - hit_used;
    for (uint32_t k = __lane_id(); k < left; k += 64)
      if (hit_base + hit_used + k < P.out.hit_capacity) hits[hit_base + hit_used + k].tag = ODW_TAG_UNUSED;
    const uint64_t at = hit_base + hit_used;
    const uint64_t in_buf = at < P.out.hit_capacity ? (P.out.hit_capacity - at < left ? P.out.hit_capacity - at : left) : 0;
    if (__lane_id() == 0 && in_buf) atomicAdd(hit_count + 1, (unsigned long long)in_buf);
  }
  if (__lane_id() == 0) {          // "full": the next append reserves a fresh block (of the new scene's segment)
    hit_state[0] = 0u; hit_state[1] = 0u; hit_state[2] = P.out.hit_block; hit_state[3] = next_scene;
  }
}

template <bool ON> struct HitBlockState {
  __device__ static __forceinline__ uint32_t* lds() {
    __shared__ uint32_t state[4 * 4 + 4 + ODW_HIST_WIN * ODW_HIST_WIN];
    return state;
  }
};
template <> struct HitBlockState<false> {
  __device__ static __forceinline__ uint32_t* lds() { return nullptr; }
};

// ------------------------------------------------------------ the kernel
#ifndef ODW_CHUNK
#define ODW_CHUNK 2048ull      // rays per hand-out unit of long launches (32 per lane); TraceParams.chunk is what a launch uses
#endif
#ifndef ODW_REFILL_MIN
#define ODW_REFILL_MIN 16      // idle lanes that trigger a refill of a partly busy wave
#endif
#ifndef ODW_WAVES_PER_SIMD
#define ODW_WAVES_PER_SIMD 4
#endif
#ifndef ODW_WAVES_PER_SIMD_BVH
#define ODW_WAVES_PER_SIMD_BVH 4
#endif
// LEAN: the scene has no grating group and no finite absorption length (the host checks): their code
// -- line_grating's chain of IEEE divisions and square roots, exp() -- is left out of the binary
// BATCH (flat kernels): scenes of one structure side by side in one launch (DeviceBatch)
template <bool BVH, bool STOCH, bool SEG, bool LEAN = false, class SPEC = NoSpec, bool BATCH = false>
__device__ __forceinline__ void trace_body(const TraceParams& P) {
  static_assert(!BATCH || (!BVH && !SEG), "batch launches: flat kernels, no segment rows");
  extern __shared__ int bvh_stack[];  // ODW_BVH_STACK x 256 ints (BVH variant only)
  // per-thread event counters live in LDS (one column per thread, ds_add_u32
  // at the event): eight fewer VGPRs across the whole ray loop
  __shared__ uint32_t cnt_lds[ODW_CNT_LDS * 256];
#pragma unroll
  for (int k = 0; k < ODW_CNT_LDS; ++k) cnt_lds[k * 256 + threadIdx.x] = 0;
#define ODW_COUNT(k) (cnt_lds[(k) * 256 + threadIdx.x] += 1u)
  const DeviceScene& sc = P.scene;
  const DeviceLimits& lim = P.lim;
  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_hdr = as_const(sc.prim_hdr);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  cf64 group_f64 = as_const(sc.group_f64);
  ci32 group_i32 = as_const(sc.group_i32);
  cf64 group_gdir = as_const(sc.group_gdir);
  cu64 seq_mask = as_const(sc.seq_mask);
  // BATCH: the scene the wave's rays belong to, and the scene of the hand-out unit it holds (wave-uniform)
  uint32_t scene = 0, unit_scene = 0;
  // Persistent waves with ray regeneration.  Rays are handed out in chunks of
  // ODW_CHUNK consecutive indices, taken from a launch-wide atomic counter.  A
  // lane whose ray has terminated takes the next index of its wave's chunk
  // (ballot + popcount prefix, no atomics) instead of idling until the slowest
  // ray of the wave is done -- in scenes like hugeArray path lengths range
  // from 1 to 100 segments, and a wave would otherwise run at the pace of its
  // longest path.  One loop iteration = one segment of every live lane.
  const uint32_t lane = __lane_id();
  uint64_t next = 0, chunk_end = 0;                        // wave-uniform
  bool alive = false;
  uint64_t i = 0;
  d3 point = mk(0, 0, 0), dir = mk(0, 0, 1);
  double power = 0;
  int seq = 0, nint = 0, medium = -1;
  int skip = -1;     // solid the ray has just left, if that solid is convex (it cannot be met again)
  int only = -1;     // solid the ray has just entered, if that solid is isolated (it is met before anything else)
  // per wave: block of hit-list slots (base lo, base hi) and how many are taken (full: none reserved yet)
  // (flat kernels only: the BVH kernels' node stacks + counters fill the 160 KB of a CU exactly at
  //  4 blocks -- 64 more bytes would cost a quarter of the occupancy)
  uint32_t* hit_lds = HitBlockState<!BVH>::lds();
  uint32_t* hist_win = BVH ? nullptr : hit_lds + 16;
  if (!BVH) {
    if (threadIdx.x < 16) hit_lds[threadIdx.x] = (threadIdx.x & 3) == 2 ? P.out.hit_block : 0u;
    for (int k = threadIdx.x; k < 4 + ODW_HIST_WIN * ODW_HIST_WIN; k += 256) hist_win[k] = 0u;
    __syncthreads();
  }
#ifdef ODW_FLAT_STATS
  // diagnostic build: s_memtime ticks per wave in refill + generation / nearest-hit search / the rest (odw_destroy prints)
  uint64_t flat_t[3] = {0, 0, 0};
  uint64_t flat_mark = __builtin_readcyclecounter();
#define ODW_FTIME(k) do { const uint64_t t_ = __builtin_readcyclecounter(); flat_t[(k)] += t_ - flat_mark; flat_mark = t_; } while (0)
#else
#define ODW_FTIME(k) do {} while (0)
#endif
  for (;;) {
    ODW_FTIME(2);
    const uint64_t idle = __ballot(!alive);
    // refill when the wave is empty or enough lanes are idle to make the
    // (divergent) generation code worth running
    if (idle && (idle == ~0ull || __popcll(idle) >= ODW_REFILL_MIN)) {
      if (next >= chunk_end) {
        // next chunk of the launch: one atomic per wave and chunk (dynamic
        // hand-out keeps every CU busy until the very end of the launch)
        unsigned long long c = 0;
        if (lane == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1)) c = atomicAdd(P.out.chunk_counter, 1ull);
        const uint64_t chunk = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)c);
        if (BATCH) {
          // unit g of the launch = unit g % chunks_per_scene of scene g / chunks_per_scene; ray indices are the scene's own
          const uint32_t g = (uint32_t)chunk, cps = P.batch.chunks_per_scene;
          if (chunk < (uint64_t)cps * P.batch.n_scenes) {
            unit_scene = g / cps;
            next = (uint64_t)(g - unit_scene * cps) * P.chunk;
            chunk_end = next + P.chunk < P.batch.rays ? next + P.chunk : P.batch.rays;
          } else {
            next = chunk_end = 0;                              // the launch has handed out everything
          }
        } else {
        next = chunk * (uint64_t)P.chunk;
        if (next > P.n_rays) next = P.n_rays;
        chunk_end = next + P.chunk < P.n_rays ? next + P.chunk : P.n_rays;
        }
      }
      uint64_t avail = next < chunk_end ? chunk_end - next : 0;
      if (BATCH && avail && unit_scene != scene) {
        // the unit belongs to another scene than the rays this wave still traces: they finish first (a wave meets
        // every scene boundary about once per launch), then the wave closes its block of the old scene's hit-list
        // segment and moves its table pointers
        if (idle != ~0ull) {
          avail = 0;
        } else {
          close_hit_block(P, hit_lds + (threadIdx.x >> 6) * 4, unit_scene);    // (closes the old scene's block, notes the new scene)
          scene = unit_scene;
          const size_t off = (size_t)scene * (size_t)P.batch.stride;
          sv.prim_f64 = as_const(sc.prim_f64 + off);
          sv.prim_hdr = as_const(sc.prim_hdr + off);
          group_f64 = as_const(sc.group_f64 + off);
          group_gdir = as_const(sc.group_gdir + off);
        }
      }
      const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
      const uint32_t want = __popcll(idle);
      const uint32_t take = want < avail ? want : (uint32_t)avail;
      if (!alive && rank < take) {
        i = next + rank;
        if (P.ray_origins) {
          // (component-major staging: the lanes of a wave take consecutive rays, so these are coalesced loads)
          point = mk(P.ray_origins[i], P.ray_origins[P.ray_stride + i], P.ray_origins[2 * P.ray_stride + i]);
          dir = mk(P.ray_dirs[i], P.ray_dirs[P.ray_stride + i], P.ray_dirs[2 * P.ray_stride + i]);
          dir = dir * (1.0 / sqrt(dot(dir, dir)));
          power = P.ray_powers ? P.ray_powers[i] : 1.0;
        } else if (BATCH && P.batch.gen_dirs) {
          // (generated once for all scenes of the batch, odw_batch_rays_kernel: the values generate_ray() returns)
          const double* gd = P.batch.gen_dirs;
          const uint64_t gs = P.batch.gen_stride;
          dir = mk(gd[i], gd[gs + i], gd[2 * gs + i]);
          if (P.batch.gen_origins) point = mk(P.batch.gen_origins[i], P.batch.gen_origins[gs + i], P.batch.gen_origins[2 * gs + i]);
          else point = mk(gd[3 * gs], gd[3 * gs + 1], gd[3 * gs + 2]);
          power = as_const(P.source)->power;
        } else {
          const RayInit r = generate_ray(P.source, P.first_ray + i, P.seed);
          point = r.point; dir = r.dir; power = r.power;
#if ODW_DOUBLE == 8
          {
            uint32_t lo = (uint32_t)(P.first_ray + i);
            asm volatile("" : "+v"(lo));
            const RayInit r2 = generate_ray(P.source, ((P.first_ray + i) & ~0xFFFFFFFFull) | lo, P.seed);
            dir = r2.dir.x == dir.x ? dir : r2.dir;
          }
#endif
        }
        // `dir` stays a unit vector: mirror() preserves length, snells_law() and
        // line_grating() return unit vectors for unit input; the reference
        // renormalises every segment (ray.py:377), a no-op up to rounding
        seq = 0; nint = 0; medium = -1; skip = -1; only = -1;
        alive = true;
      }
      next += take;
      if (take == 0 && idle == ~0ull) break;               // nothing live, nothing left (BATCH: a waiting unit is taken above once all lanes idle)
    }
    ODW_FTIME(0);
    if (alive) {
      if (nint >= lim.max_intersections) {
        ODW_COUNT(ODW_CNT_CAPPED);
        alive = false;
      } else {
      ++nint;
      uint64_t mask = sc.all_mask;
      if (sc.seq_enabled) mask = (seq < sc.seq_len) ? seq_mask[seq] : 0ull;
      mask &= ~sc.ignore_mask;
      double t_hit;
      int face;
      const int prim = nearest<BVH, SPEC>(sc, sv, lim, point, dir, medium, mask, t_hit, face,
                                          bvh_stack + threadIdx.x, skip, BVH ? -1 : only);
      // A ray that has entered an isolated solid within distTol beyond one of its edges can pass it by: then, and
      // only then, the solid's own primitives yield nothing -- the segment is done again with every primitive, in
      // the next trip of the loop (no second copy of the search, no state kept across one)
      if (!BVH && prim < 0 && only >= 0) {
        only = -1;
        --nint;
        continue;
      }
      ODW_FTIME(1);
      if (SEG)   // (p1, p2), power at p1, medium of the segment (ray.py:104-117)
        record_segment(P.out.segs, P.out.seg_capacity, P.out.seg_count, P.first_ray + i, nint - 1, medium, point,
                       point + dir * (prim < 0 ? lim.max_ray_length : t_hit), power);
      if (prim < 0) {
        ODW_COUNT(ODW_CNT_ESCAPED);
        alive = false;
      } else {
      point = point + dir * t_hit;
      // absorption along the traversed medium (ray.py:120-125, assignment)
      if (!LEAN && medium >= 0) {
        const double L = group_f64[4 * medium + 2];
        if (L == 0) power = 0;
        else if (L < INFINITY) power = exp(-t_hit / L);
      }
      uint32_t* cnt = cnt_lds + threadIdx.x;
      uint32_t* hit_state = hit_lds + (threadIdx.x >> 6) * 4;
      if constexpr (SPEC::enabled) {
        spec_hits<STOCH, LEAN, SPEC>(P, sv, group_f64, group_i32, group_gdir, prim, face, P.first_ray + i, nint, cnt, hit_state,
                              hist_win, point, dir, power, medium, seq, skip, only, alive, __make_integer_seq<IndexList, int, SPEC::N>{});
      } else {
      cf64 pf = sv.prim_f64 + (size_t)prim * 16;
      ci32 pi = sv.prim_i32 + 4 * prim;
      // getNormal (ray.py:455-480): outward normal -> along the travel direction
      d3 n;
      if (BVH && pi[0] == ODW_PRIM_TRIANGLE) {
        n = tri_normal(pf, sc.tri_nrm ? sc.tri_nrm + (size_t)prim * 9 : nullptr, point);
        if (pi[2] & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
      } else {
        n = face_normal<BVH>(pi[0], pf + 12, face, xf_point(pf, point));
        if (pi[2] & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
        n = xf_vec_t(pf, n);
      }
      const bool entering = dot(dir, n) < 0;
      if (entering) n = n * -1.0;
      const int g = pi[1];
      interact<BVH, STOCH, LEAN>(P, group_f64, group_i32, group_gdir, g, group_i32[4 * g], group_i32[4 * g + 1] != 0, n,
                                 entering, P.first_ray + i, nint, cnt, hit_state, hist_win, point, dir, power, medium, seq, alive);
      // outward normal of the solid = n against the travel direction when entering; for a facet of a convex
      // tessellated solid the FACET's own normal decides (the interpolated one of smooth shading can point out of
      // the solid where the ray still runs into it)
      double out = entering ? -dot(dir, n) : dot(dir, n);
      if (BVH && pi[0] == ODW_PRIM_TRIANGLE) {
        out = dot(dir, mk(pf[9], pf[10], pf[11]));
        if (pi[2] & ODW_FLAG_FLIP_NORMAL) out = -out;
      }
      skip = ((pi[2] & ODW_FLAG_CONVEX) && out > 0) ? (pi[2] >> ODW_SOLID_SHIFT) : -1;
      only = (!BVH && (pi[2] & ODW_FLAG_ISOLATED) && out < 0) ? (pi[2] >> ODW_SOLID_SHIFT) : -1;
      }
      if (alive && power < lim.power_tol) { ODW_COUNT(ODW_CNT_DIED); alive = false; }
      }
      }
      if (!alive) {
        cnt_lds[ODW_CNT_SEGMENTS * 256 + threadIdx.x] += (uint32_t)nint;
        ODW_COUNT(ODW_CNT_TRACED_RAYS);
      }
    }
  }
#ifdef ODW_FLAT_STATS
  if (!BVH && lane == 0 && P.dbg)
    for (int k = 0; k < 3; ++k) atomicAdd(P.dbg + 28 + k, (unsigned long long)flat_t[k]);
#endif
  // slots of the last block this wave never filled
  if (!BVH) close_hit_block(P, hit_lds + (threadIdx.x >> 6) * 4, BATCH ? scene : 0u);
  // the block's histogram window joins the histogram (every wave of the block has finished its rays)
  if (!BVH) {
    __syncthreads();
    if (hist_win[0] != 0u) {
      cdetector det = as_const(opaque(P.det));
      const uint32_t x0 = hist_win[0] - 1u, y0 = hist_win[1];
      const uint32_t nx = (uint32_t)det->nx, ny = (uint32_t)det->ny;
      for (int k = threadIdx.x; k < ODW_HIST_WIN * ODW_HIST_WIN; k += 256) {
        const uint32_t c = hist_win[4 + k];
        const uint32_t x = x0 + (uint32_t)k / ODW_HIST_WIN, y = y0 + (uint32_t)k % ODW_HIST_WIN;
        if (c && x < nx && y < ny) atomicAdd(P.out.hist + ((size_t)x * ny + y), (unsigned long long)c);
      }
    }
  }
  // counters: wave reduction, then one atomic per BLOCK and counter (one per wave put 4e4 memory-side atomics on
  // eight addresses at the end of every launch: 0.07 ms of a 1e7-ray launch's 1.2)
  // (a wave's sums go into the first column of its own quarter of the counter table: no LDS beyond what the kernel
  //  has -- the BVH kernels' node stacks + this table fill the CU's 160 KB exactly at four blocks)
#pragma unroll
  for (int k = 0; k < ODW_CNT_LDS; ++k) {
    const uint32_t s = wave_sum(cnt_lds[k * 256 + threadIdx.x]);
    if (__lane_id() == 0) cnt_lds[k * 256 + (threadIdx.x & ~63u)] = s;
  }
  __syncthreads();
  if (threadIdx.x < ODW_CNT_LDS) {
    const uint32_t* col = cnt_lds + threadIdx.x * 256;
    const uint32_t s = col[0] + col[64] + col[128] + col[192];
    if (s) atomicAdd(P.out.counters + threadIdx.x, (unsigned long long)s);
  }
}

template <bool BVH, bool STOCH, bool SEG, bool LEAN = false, bool BATCH = false>
__global__ __launch_bounds__(256, BVH ? ODW_WAVES_PER_SIMD_BVH : ODW_WAVES_PER_SIMD) void odw_trace_kernel(const TraceParams P) {
  trace_body<BVH, STOCH, SEG, LEAN, NoSpec, BATCH>(P);
}

#ifdef ODW_SPEC_HEADER
// the scene-compiled flat kernel: ODW_SPEC_HEADER defines `struct Spec` (written by the host library
// from the uploaded scene, odw_spec.hip: spec_text), ODW_SPEC_LEAN and ODW_SPEC_STOCH
#include ODW_SPEC_HEADER
#ifndef ODW_SPEC_WAVES
#define ODW_SPEC_WAVES 4
#endif
#ifndef ODW_SPEC_STOCH
#define ODW_SPEC_STOCH false
#endif
#ifndef ODW_SPEC_BATCH
#define ODW_SPEC_BATCH false
#endif
extern "C" __global__ __launch_bounds__(256, ODW_SPEC_WAVES) void odw_spec_kernel(const TraceParams P) {
  trace_body<false, ODW_SPEC_STOCH, false, ODW_SPEC_LEAN, Spec, ODW_SPEC_BATCH>(P);
}
#endif

// sampler only: theta-or-radius and phi per ray (diagnostics / parity tests)
__global__ __launch_bounds__(256) void odw_sample_kernel(const DeviceSource* sp, uint64_t first,
                                                         uint64_t n, uint64_t seed,
                                                         double* __restrict__ t_out,
                                                         double* __restrict__ phi_out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double up, ut, t, phi;
    ray_uniforms(first + i, seed, up, ut);
    sample_source(as_const(sp), up, ut, t, phi);
    t_out[i] = t;
    phi_out[i] = phi;
  }
}


// --------------------------------------------------------- surface source
// SurfaceSourceProxy._generateRays(mode='true') (surface_source.py:519-553)
#define ODW_EMIT_MAX_ATTEMPTS 4096
__device__ __forceinline__ void philox_pair(uint64_t ray, uint64_t seed, uint32_t c2, uint32_t c3,
                                            double& a, double& b) {
  uint32_t c0 = (uint32_t)ray, c1 = (uint32_t)(ray >> 32);
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
  a = u53(c0, c1);
  b = u53(c2, c3);
}

// point, outward normal and tangent (d/du of the face's parametrisation) of
// face `face` of a primitive at face coordinates (ua, ub) in [0,1)^2; returns
// the acceptance probability of the point (torus: area element / its maximum)
__device__ __forceinline__ double face_point(int type, const double* par, int face, double ua, double ub,
                                             d3& p, d3& n, d3& t) {
  const double two_pi = 6.283185307179586;
  if (type == ODW_PRIM_BOX) {
    const int a = face >> 1, b1 = (a + 1) % 3, b2 = (a + 2) % 3;
    double c[3];
    c[a] = (face & 1) ? par[a] : 0.0;
    c[b1] = ua * par[b1];
    c[b2] = ub * par[b2];
    p = mk(c[0], c[1], c[2]);
    const double s = (face & 1) ? 1.0 : -1.0;
    n = mk(a == 0 ? s : 0.0, a == 1 ? s : 0.0, a == 2 ? s : 0.0);
    t = mk(b1 == 0 ? 1.0 : 0.0, b1 == 1 ? 1.0 : 0.0, b1 == 2 ? 1.0 : 0.0);
    return 1.0;
  }
  double sa, ca;
  sincos_bounded(two_pi * ua, sa, ca);
  if (type == ODW_PRIM_SPHERE) {
    const double z = 2.0 * ub - 1.0, r = sqrt(fmax(0.0, 1.0 - z * z));
    n = mk(r * ca, r * sa, z);
    p = n * par[0];
    t = mk(-sa, ca, 0.0);
    return 1.0;
  }
  if (type == ODW_PRIM_TORUS) {
    double sv, cv;
    sincos_bounded(two_pi * ub, sv, cv);
    const double rho = par[0] + par[1] * cv;
    p = mk(rho * ca, rho * sa, par[1] * sv);
    n = mk(cv * ca, cv * sa, sv);
    t = mk(-sa, ca, 0.0);
    return rho / (par[0] + par[1]);
  }
  // cylinder / cone
  const double r1 = par[0], r2 = (type == ODW_PRIM_CONE) ? par[1] : par[0];
  const double h = (type == ODW_PRIM_CONE) ? par[2] : par[1];
  if (face == 0) {
    double z;                               // area element ~ radius(z)
    if (r1 == r2) z = h * ub;
    else z = h * (sqrt(r1 * r1 + ub * (r2 * r2 - r1 * r1)) - r1) / (r2 - r1);
    const double k = (r2 - r1) / h, r = r1 + k * z;
    p = mk(r * ca, r * sa, z);
    const double inv = 1.0 / sqrt(1.0 + k * k);
    n = mk(ca * inv, sa * inv, -k * inv);
    t = mk(-sa, ca, 0.0);
    return 1.0;
  }
  const double rr = (face == 1 ? r1 : r2) * sqrt(ub);
  p = mk(rr * ca, rr * sa, face == 1 ? 0.0 : h);
  n = mk(0.0, 0.0, face == 1 ? -1.0 : 1.0);
  t = mk(1.0, 0.0, 0.0);
  return 1.0;
}

// cs / rs: stride of a component / of a ray in the output (3 x n component-major for the trace kernels: cs = n,
// rs = 1; n x 3 for odw_generate_rays: cs = 1, rs = 3)
__global__ __launch_bounds__(256) void odw_emit_kernel(const DeviceEmitter E, uint64_t first, uint64_t n, uint64_t seed,
                                                       double* __restrict__ origins, double* __restrict__ dirs, uint64_t cs,
                                                       uint64_t rs) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t ray = first + i;
    d3 gp = mk(0, 0, 0), gn = mk(0, 0, 1), gt = mk(1, 0, 0);
    for (uint32_t attempt = 0; attempt < ODW_EMIT_MAX_ATTEMPTS; ++attempt) {
      double u_face, u_acc, ua, ub;
      philox_pair(ray, seed, attempt, 3u, u_face, u_acc);
      philox_pair(ray, seed, attempt, 4u, ua, ub);
      int f = 0, hi = E.n_faces - 1;          // largest f with cdf[f] <= u_face
      while (f < hi) {
        const int mid = (f + hi + 1) >> 1;
        if (u_face >= E.face_cdf[mid]) f = mid; else hi = mid - 1;
      }
      const int prim = E.face_i32[2 * f], face = E.face_i32[2 * f + 1];
      const double* pf = E.prim_f64 + (size_t)prim * 16;
      const int32_t* pi = E.prim_i32 + 4 * prim;
      if (pi[0] == ODW_PRIM_TRIANGLE) {       // a facet of a tessellated face: global coordinates, no conditions
#pragma clang fp contract(off)
        const d3 v0 = mk(pf[0], pf[1], pf[2]);
        const d3 e1 = mk(pf[3], pf[4], pf[5]) - v0, e2 = mk(pf[6], pf[7], pf[8]) - v0;
        double a = ua, b = ub;
        if (a + b > 1.0) { a = 1.0 - a; b = 1.0 - b; }
        gp = v0 + (e1 * a + e2 * b);
        const d3 fn = cross(e1, e2);
        gn = fn * (1.0 / sqrt(dot(fn, fn)));
        if (E.tri_nrm) {
          const double* vn = E.tri_nrm + (size_t)prim * 9;
          const d3 mix = mk(vn[0], vn[1], vn[2]) * (1.0 - a - b) + (mk(vn[3], vn[4], vn[5]) * a + mk(vn[6], vn[7], vn[8]) * b);
          gn = mix * (1.0 / sqrt(dot(mix, mix)));
        }
        if (pi[1] & ODW_FLAG_FLIP_NORMAL) gn = gn * -1.0;
        gt = e1 - gn * dot(e1, gn);
        gt = gt * (1.0 / sqrt(dot(gt, gt)));
        break;
      }
      d3 p, nl, tl;
      const double accept = face_point(pi[0], pf + 12, face, ua, ub, p, nl, tl);
      if (u_acc >= accept) continue;
      if (pi[1] & ODW_FLAG_FLIP_NORMAL) nl = nl * -1.0;
      // local -> global: x = R^T (p - t)
      const d3 q = mk(p.x - pf[3], p.y - pf[7], p.z - pf[11]);
      gp = xf_vec_t(pf, q);
      bool ok = true;
      for (int c = pi[2]; c < pi[2] + pi[3] && ok; ++c) {
        const int cw = E.cond_i32[c];
        const int qp = cw & 0x7fffffff;
        const double* of = E.prim_f64 + (size_t)qp * 16;
        const double sd = prim_sdist<true>(E.prim_i32[4 * qp], of + 12, xf_point(of, gp));
        if (cw < 0) { if (sd > E.dist_tol) ok = false; }
        else { if (sd < -E.dist_tol) ok = false; }
      }
      if (!ok) continue;
      gn = xf_vec_t(pf, nl);
      gt = xf_vec_t(pf, tl);
      break;
    }
    double u_t, u_phi;
    philox_pair(ray, seed, 0u, 5u, u_t, u_phi);
    const int k = (int)(u_t * (double)E.n_guide);
    const double theta = inv_cdf(E.t_tab, E.t_guide[k], min(E.t_guide[k + 1] + 1, E.n_t_knots - 1), u_t);
    const double phi = 6.283185307179586 * u_phi;
    d3 d = rotate(gn, phi, rotate(gt, theta, gn));
    d = d * (1.0 / sqrt(dot(d, d)));
    origins[i * rs] = gp.x; origins[cs + i * rs] = gp.y; origins[2 * cs + i * rs] = gp.z;
    dirs[i * rs] = d.x; dirs[cs + i * rs] = d.y; dirs[2 * cs + i * rs] = d.z;
  }
}

// n x 3 (the caller's layout) -> 3 x n (the trace kernels'), both arrays in one pass
__global__ __launch_bounds__(256) void odw_rays_to_components_kernel(const double* __restrict__ o_in, const double* __restrict__ d_in,
                                                                     uint64_t n, double* __restrict__ o_out,
                                                                     double* __restrict__ d_out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n; i += stride) {
    const uint64_t ray = i / 3, c = i - 3 * ray;           // (reads are contiguous, writes three interleaved streams)
    o_out[c * n + ray] = o_in[i];
    d_out[c * n + ray] = d_in[i];
  }
}

// the rays of a batch launch, generated once for all its scenes (DeviceBatch.gen_dirs): exactly generate_ray()'s values
__global__ __launch_bounds__(256) void odw_batch_rays_kernel(const DeviceSource* sp, uint64_t first, uint64_t n, uint64_t seed,
                                                             double* __restrict__ dirs, double* __restrict__ origins,
                                                             uint64_t stride) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const RayInit r = generate_ray(sp, first + i, seed);
    dirs[i] = r.dir.x; dirs[stride + i] = r.dir.y; dirs[2 * stride + i] = r.dir.z;
    if (origins) { origins[i] = r.point.x; origins[stride + i] = r.point.y; origins[2 * stride + i] = r.point.z; }
    else if (i == 0) { dirs[3 * stride] = r.point.x; dirs[3 * stride + 1] = r.point.y; dirs[3 * stride + 2] = r.point.z; }
  }
}

// point source: initial conditions only (odw_generate_rays)
__global__ __launch_bounds__(256) void odw_make_rays_kernel(const DeviceSource* sp, uint64_t first, uint64_t n,
                                                            uint64_t seed, double* __restrict__ origins,
                                                            double* __restrict__ dirs) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double up, ut, t, phi;
    d3 o, d;
    ray_uniforms(first + i, seed, up, ut);
    sample_source(as_const(sp), up, ut, t, phi);
    make_ray(as_const(sp), t, phi, o, d);
    origins[3 * i] = o.x; origins[3 * i + 1] = o.y; origins[3 * i + 2] = o.z;
    dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
  }
}

}  // namespace odw
