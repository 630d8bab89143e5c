// odw_mesh.hip -- trace kernel for scenes with facets (tessellated shapes, STL, BRep faces that are not quadrics):
// an eight-wide tree with quantised child boxes (odw_capi.hip: WideBvh) walked by a per-lane state machine whose
// loop iteration is ONE NODE or ONE GROUP OF LEAVES, not one segment -- the construction of odw_grid.hip carried
// over to a tree.
//
// Why (ball lens as 6.5e4 / 1e6 facets, profiles/r03/r03f_mesh*_pmc.json): in odw_trace_kernel<true, ...> every
// lane of a wave finishes its traversal before any lane goes on -- 17 of 64 lanes did work per issued VALU
// instruction, 72 % of the wave cycles waited on a chain of ~40 dependent node fetches per segment --, and every
// facet of every leaf met went through the float64 Moeller-Trumbore test with its 128-byte record.  Here:
//   walk      every iteration, every walking lane visits one node: one 80-byte fetch holds the boxes of eight
//             children (8-bit offsets from the node's corner, a power-of-two scale per axis), tested in float32 with
//             the slack of the binary kernels; the children that are hit travel as ONE stack entry (first child,
//             slot bits), visited in the order `slot XOR ray octant` -- front to back without a sort.  A third of
//             the dependent fetches of the binary tree, a stack of 12 entries instead of 32.
//   leaves    a lane that hits leaf children stops walking.  When few lanes still walk, the waiting lanes test their
//             facets together: a float32 filter first (64-byte leaf records: the facet relative to the centre of
//             its group, the ray moved to the point next to that centre -- all magnitudes are those of the group,
//             the test is conservative with bounds that scale with it); only the facets that pass -- one or two per
//             leaf -- take the float64 test of odw_kernels.hip (intersect_tri: tolerance rim, face-outline rule), so
//             the hit itself is the number the binary kernels and the oracle compute.  Analytic primitives listed in
//             a leaf (a screen behind the mesh) skip the filter.
//   interact  lanes whose stack is empty: nearest-hit selection, normal (interpolated for facets), Snell / mirror
//             / grating, hit row, next segment -- or a new ray from the wave's ring (filled 64 rays at a time).
// No segment rows here (RecordRays launches of a handful of rays): those keep odw_trace_kernel<true, ...>.
// Same rules as nearest<>() (ray.py:290-452): every candidate goes through consider(), subtrees are culled
// against nearest + 2 distTol.
#include "odw_device.h"

namespace odw {

#ifndef ODW_MESH_STEP_MIN
#define ODW_MESH_STEP_MIN 16     // keep walking while at least this many lanes of the wave walk and others wait
                                 // (round 3, rays in index order, 1e6 facets: 8: 26.1 ms, 16: 27.0, 28: 29.8 per 1e7 rays;
                                 //  round 4, rays handed out in sorted order -- the lanes of a wave take the same turns --:
                                 //  8: 13.0, 12: 11.8, 16: 11.55, 24: 11.6 at 1e6 facets, 7.9 / 7.65 / 7.45 at 6.5e4)
#endif
#ifndef ODW_MESH_WAVES
#define ODW_MESH_WAVES 3         // waves per SIMD the register allocation aims at (LDS per block: stacks 24 KB + rings 14 KB;
                                 // 4: the hot loops spill -- 16.5 against 11.55 ms)
#endif
#ifndef ODW_MESH_CAND_TRIPS
#define ODW_MESH_CAND_TRIPS 0    // trips of four records per lane and leaf round; 0: all of the visit's candidates at once.
                                 // (round 3: 0: 13.1 / 18.3 / 25.1 ms per 1e7 rays at 4e3 / 6.5e4 / 1e6 facets, 3: 12.9 / 16.9 /
                                 //  24.4 -- a lane with 40 candidates held the lanes with 8; with sorted rays the lanes of a wave
                                 //  hold about the same number: 0: 7.65 / 11.55, 3: 7.95 / 12.4, 4: 7.93 / 12.0, 6: 8.36 / 12.4)
#endif
#ifndef ODW_MESH_CONES
#define ODW_MESH_CONES 1         // rays inside a strictly convex tessellated solid drop the slots whose facets all face them
                                 // (1e7 rays, 4e3 / 6.5e4 / 1e6 facets, ms: 4.13 / 4.98 / 7.47 without, 3.77 / 4.34 / 5.96 with:
                                 // node visits per segment 11.7 -> 8.9, candidate facets 18 -> 12.8)
#endif
#ifndef ODW_MESH_INTERACT_MIN
#define ODW_MESH_INTERACT_MIN 64   // sorted hand-out order: lanes done with the tree before the wave interacts (64: all of them)
#endif
#define ODW_MESH_THREADS 256
#define ODW_MESH_BLOCK_WAVES (ODW_MESH_THREADS / 64)
#define ODW_MESH_WAVE_WORDS 32   // per wave: event counters (0..7), diagnostics (8..27), hit-block state (28..31)
#define ODW_MESH_RING 64
#define ODW_MESH_RING_DOUBLES (ODW_MESH_RING * 7)   // per ray: origin, direction, its number within the launch
#define ODW_MESH_STACK 12        // entries (two words each) per lane: one per level of the wide tree (kWideMaxDepth + 1)
#define ODW_WIDE_WORDS 32        // words per node (odw_capi.hip: kWideWords)

// leaf record (64 bytes): words 0..2 v0 - centre of its group, 3..5 e1, 6..8 e2 (float32), 9 group | solid << 8 |
// (not a facet) << 31, 10 largest barycentric slack per unit of tolerance, 11 error scale: 4e-6 x max(|e1|_1,
// |e2|_1), 12 primitive, 13..15 the centre (the same in every record of a node's leaves)
#define ODW_LEAF_WORDS 16

#ifdef ODW_MESH_STATS
#define ODW_MSTAT(k, mask_)                                                                           \
  do {                                                                                                \
    const unsigned long long m_ = (mask_);                                                            \
    if (m_ && (int)__lane_id() == __ffsll(m_) - 1) { wave_cnt[8 + 2 * (k)] += 1u; wave_cnt[9 + 2 * (k)] += (uint32_t)__popcll(m_); } \
  } while (0)
// time per phase: s_memtime ticks between marks, kept per wave (uniform), added up at the end
#define ODW_MTIME(k) do { const uint64_t t_ = __builtin_readcyclecounter(); phase_t[(k)] += t_ - t_mark; t_mark = t_; } while (0)
// a per-lane quantity added up (runs: the lanes that report, lanes: the sum)
#define ODW_MSUM(k, n_) do { atomicAdd(&wave_cnt[8 + 2 * (k)], 1u); atomicAdd(&wave_cnt[9 + 2 * (k)], (uint32_t)(n_)); } while (0)
#else
#define ODW_MSTAT(k, mask_) do {} while (0)
#define ODW_MTIME(k) do {} while (0)
#define ODW_MSUM(k, n_) do {} while (0)
#endif
#define ODW_MCOUNT(k) atomicAdd(&wave_cnt[(k)], 1u)

// what a lane does at the end of a segment (ray.py:120-268): absorption along the segment, normal, hit row, the new
// direction, the solid a convex facet lets the ray leave.  A function of its own so that it can be kept out of line
// (ODW_MESH_INTERACT_INLINE=0; kargs: the kernel's argument segment, as record_hit_flat) -- measured in round 5
// (1e7 rays, 4e3 / 6.5e4 / 1e6 facets, ms): inlined 4.11 / 4.93 / 7.36 with 47 spilled registers, out of line 4.47 / 5.28 /
// 7.70 with 22 (the call moves ~50 registers per segment and lane, the spills it saves were not in the hot loops); out of
// line at four waves per SIMD (ODW_MESH_WAVES=4, still 96 spilled) 5.47 / 6.45 / 9.12.
#ifndef ODW_MESH_INTERACT_INLINE
#define ODW_MESH_INTERACT_INLINE 1
#endif
struct MeshRay { d3 point, dir; double power; int medium, seq, skip, inside; bool alive; };
template <bool STOCH>
#if ODW_MESH_INTERACT_INLINE
__device__ __forceinline__
#else
__device__ __noinline__
#endif
MeshRay mesh_interact(ckargs kargs, d3 point, d3 dir, double power, int medium, int seq, int nint, uint64_t i, double t_hit, int prim,
                      int face, uint32_t* wave_cnt, volatile uint32_t* hit_state, const double* group_f64, const int32_t* group_i32,
                      const double* group_gdir) {
  const uint64_t a_ = (uint64_t)(uintptr_t)kargs;
  const uint64_t u_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a_ >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a_);
  const TraceParams ODW_CONST& P = *(ckargs)(uintptr_t)u_;
  const DeviceScene ODW_CONST& sc = P.scene;
  const DeviceLimits ODW_CONST& lim = P.lim;
  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_hdr = as_const(sc.prim_hdr);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  bool alive = true;
  int skip = -1, inside = -1;
  {
        cf64 pf = sv.prim_f64 + (size_t)prim * 16;
        ci32 pi = sv.prim_i32 + 4 * prim;
        point = point + dir * t_hit;
        if (medium >= 0) {                                  // ray.py:120-125 (assignment)
          const double L = group_f64[4 * medium + 2];
          if (L == 0) power = 0;
          else if (L < INFINITY) power = exp(-t_hit / L);
        }
        d3 n;
        if (pi[0] == ODW_PRIM_TRIANGLE) {
          n = tri_normal(pf, sc.tri_nrm ? sc.tri_nrm + (size_t)prim * 9 : nullptr, point);
          if (pi[2] & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
        } else {
          n = face_normal<true>(pi[0], pf + 12, face, xf_point(pf, point));
          if (pi[2] & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
          n = xf_vec_t(pf, n);
        }
        const bool entering = dot(dir, n) < 0;
        if (entering) n = n * -1.0;
        const int g = pi[1];
        const int gtype = group_i32[4 * g];
        if (group_i32[4 * g + 1]) {
          ODW_MCOUNT(ODW_CNT_RECORDED_HITS);
          record_hit<true, 1, true>(P, P.first_ray + i, g, point, dir, power, entering, wave_cnt, hit_state);
        }
        if (gtype == ODW_OPT_MIRROR) {
          const d3 ideal = mirror(dir, n);
          if (STOCH) dir = scatter(P.samplers, P.group_sampler[2 * g], P.group_sampler[2 * g + 1], P.first_ray + i, P.seed,
                                   (uint32_t)nint, dir, ideal, n, 1.0);
          else dir = ideal;
          power *= group_f64[4 * g + 1];
          ++seq;
        } else if (gtype == ODW_OPT_LENS) {
          const double n1 = (medium >= 0) ? group_f64[4 * medium] : 1.0;
          double n2 = 1.0;
          if (entering) { medium = g; n2 = group_f64[4 * g]; }
          bool tir;
          const d3 ideal = snells_law(dir, n1, n2, n, tir);
          if (STOCH) dir = scatter(P.samplers, P.group_sampler[2 * g], P.group_sampler[2 * g + 1], P.first_ray + i, P.seed,
                                   (uint32_t)nint, dir, ideal, n, tir ? -1.0 : n1 / n2);
          else dir = ideal;
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
              dir = line_grating(dir, nn, nn, n, P.wavelength, order, lpm, gd, false);
              ++seq;
            }
          } else if (entering) {
            if (medium >= 0) {
              atomicAdd(P.out.counters + ODW_CNT_GRATING_IN_MEDIUM, 1ull);     // a ValueError of the reference
              ODW_MCOUNT(ODW_CNT_DIED);
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
        {
          // (a facet of a convex tessellated solid: the facet's own normal decides, as in odw_trace_kernel)
          double out = entering ? -dot(dir, n) : dot(dir, n);
          if (pi[0] == ODW_PRIM_TRIANGLE) {
            out = dot(dir, mk(pf[9], pf[10], pf[11]));
            if (pi[2] & ODW_FLAG_FLIP_NORMAL) out = -out;
          }
          skip = ((pi[2] & ODW_FLAG_CONVEX) && out > 0) ? (pi[2] >> ODW_SOLID_SHIFT) : -1;
          // the ray goes on INSIDE a convex tessellated solid, from a facet whose edges are all closed (the point is on
          // the facet up to rounding, not up to the tolerance): the walk drops what the ray could only meet from outside
          // (normal cones, odw_capi.hip: WideBvh::cone_word)
          inside = (pi[0] == ODW_PRIM_TRIANGLE && (pi[2] & ODW_FLAG_CONVEX) && (pi[2] & ODW_FLAG_STRICTLY_CONVEX) && out < 0 &&
                    pf[12] < 0 && pf[13] < 0 && pf[14] < 0)
                       ? (pi[2] >> ODW_SOLID_SHIFT) : -1;
        }
        if (alive && power < lim.power_tol) { ODW_MCOUNT(ODW_CNT_DIED); alive = false; }
  }
  MeshRay r;
  r.point = point; r.dir = dir; r.power = power; r.medium = medium; r.seq = seq; r.skip = skip; r.inside = inside; r.alive = alive;
  return r;
}

// an analytic primitive listed in a leaf (a screen behind the mesh), out of line for the same reason: intersect_prim<>
// holds every kind of primitive, the quartic of the torus included
struct MeshBest { Best any, oth; };
#if ODW_MESH_INTERACT_INLINE
__device__ __forceinline__
#else
__device__ __noinline__
#endif
MeshBest mesh_intersect_prim(ckargs kargs, d3 start, d3 dn, double tol, double tmax, int medium, Best any, Best oth, int p) {
  const uint64_t a_ = (uint64_t)(uintptr_t)kargs;
  const uint64_t u_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a_ >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a_);
  const DeviceScene ODW_CONST& sc = ((ckargs)(uintptr_t)u_)->scene;
  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_hdr = as_const(sc.prim_hdr);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  Query q;
  q.start = start; q.dn = dn; q.tol = tol; q.tmax = tmax; q.medium = medium; q.any = any; q.oth = oth;
  ci32 pi = sv.prim_i32 + 4 * p;
  intersect_prim<true>(sv, q, p, pi[0], pi[1], pi[2], pi[3]);
  MeshBest r;
  r.any = q.any; r.oth = q.oth;
  return r;
}

// STOCH: the scene has stochastic surfaces (scatter() after the ideal mirror / Snell direction, as in interact<>)
template <bool STOCH>
__global__ __launch_bounds__(ODW_MESH_THREADS, ODW_MESH_WAVES) void odw_mesh_kernel(const TraceParams P) {
  extern __shared__ double mesh_lds[];
  const DeviceScene& sc = P.scene;
  const DeviceLimits& lim = P.lim;
  // ---- LDS image: node stacks [ODW_MESH_STACK][256] x 2 words | per-wave words | ray rings ----
  uint2* stack = reinterpret_cast<uint2*>(mesh_lds) + threadIdx.x;
  uint32_t* lds32 = reinterpret_cast<uint32_t*>(mesh_lds);
  const int word_off = 2 * ODW_MESH_STACK * ODW_MESH_THREADS;
  const int ring_off = (word_off + ODW_MESH_BLOCK_WAVES * ODW_MESH_WAVE_WORDS) / 2;     // doubles
  for (int k = threadIdx.x; k < ODW_MESH_BLOCK_WAVES * ODW_MESH_WAVE_WORDS; k += ODW_MESH_THREADS)
    lds32[word_off + k] = (k % ODW_MESH_WAVE_WORDS) == 30 ? P.out.hit_block : 0u;        // hit-block state: base lo, hi, used, -
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  uint32_t* wave_cnt = lds32 + word_off + wave * ODW_MESH_WAVE_WORDS;
  volatile uint32_t* hit_state = wave_cnt + 28;
  double* ring = mesh_lds + ring_off + wave * ODW_MESH_RING_DOUBLES;

  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_hdr = as_const(sc.prim_hdr);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  // the groups' tables, read at every interaction with a per-lane index: from LDS (as in the grid kernel)
  __shared__ double group_f64[64 * 4];
  __shared__ int32_t group_i32[64 * 4];
  __shared__ double group_gdir[64 * 3];
  {
    cf64 gf = as_const(sc.group_f64);
    ci32 gi = as_const(sc.group_i32);
    cf64 gd = as_const(sc.group_gdir);
    for (int k = threadIdx.x; k < 64 * 4; k += ODW_MESH_THREADS) { group_f64[k] = gf[k]; group_i32[k] = gi[k]; }
    for (int k = threadIdx.x; k < 64 * 3; k += ODW_MESH_THREADS) group_gdir[k] = gd[k];
  }
  __syncthreads();
  cu64 seq_mask = as_const(sc.seq_mask);
  typedef const float ODW_CONST* cf32;
  typedef const uint32_t ODW_CONST* cu32;
  typedef float vf4 __attribute__((ext_vector_type(4)));
  typedef uint32_t vu4 __attribute__((ext_vector_type(4)));
  cu32 nodes = (cu32)(uintptr_t)sc.bvh_wide;
  cf32 leaves = (cf32)(uintptr_t)sc.bvh_leaf;
  const uint32_t lane = __lane_id();
  uint64_t next = 0, chunk_end = 0;                        // wave-uniform: the wave's chunk of the launch
  uint64_t ring_base = 0;
  uint32_t ring_n = 0;
  bool drained = false;
  // lane states: !alive | fresh (segment to set up) | walking (cur = node) | pending (leaf children to test) |
  // done with the tree (alive && !fresh && !walking && !pending): interaction
  bool alive = false, fresh = false, walking = false, pending = false;
  uint64_t i = 0;
  Query q;
  q.tol = lim.dist_tol; q.tmax = lim.max_ray_length + lim.dist_tol;
  q.start = mk(0, 0, 0); q.dn = mk(0, 0, 1); q.medium = -1;
  q.any.t = INFINITY; q.any.prim = 0x7fffffff; q.any.face = 0x7fffffff;
  q.oth = q.any;
  d3& point = q.start;
  d3& dir = q.dn;
  int& medium = q.medium;
  double power = 0;
  int seq = 0, nint = 0, skip = -1, inside = -1;
  uint64_t mask = 0;
  // the walk: float32 ray (origin moved along the ray by tsh, 1 / direction, octant: bit a = direction a > 0),
  // cut-off (relative to the moved origin), node, stack height; the leaf children that wait for their test
  float ofx = 0, ofy = 0, ofz = 0, ivx = 0, ivy = 0, ivz = 0, cutf = 0, tsh = 0, dfx = 0, dfy = 0, dfz = 0;
  uint32_t oct = 0;
  int cur = -1, sp = 0, dq = 0;
  uint32_t lbase = 0, lcounts = 0, lhits = 0;      // (or, between the rounds of one visit: the candidates left, low | high word)
  bool have_cand = false;
  const float tolf = (float)lim.dist_tol * 1.000001f;
  const uint32_t interact_min = P.ray_order ? (uint32_t)ODW_MESH_INTERACT_MIN : 1u;      // (wave-uniform)

  // which of the hit children comes first: the set bit whose slot XOR octant is largest
#define ODW_MESH_FIRST(hits_, slot_)                                                         \
  do {                                                                                       \
    uint32_t m_ = (hits_);                                                                   \
    m_ = (oct & 1u) ? (((m_ & 0x55u) << 1) | ((m_ & 0xAAu) >> 1)) : m_;                      \
    m_ = (oct & 2u) ? (((m_ & 0x33u) << 2) | ((m_ & 0xCCu) >> 2)) : m_;                      \
    m_ = (oct & 4u) ? (((m_ & 0x0Fu) << 4) | ((m_ & 0xF0u) >> 4)) : m_;                      \
    (slot_) = (uint32_t)(31 - __clz((int)m_)) ^ oct;                                         \
  } while (0)
  // the walk goes on with the next child of the group on top of the stack (first inner child, inner slots << 8 |
  // slots still to visit), or the traversal is over
#define ODW_MESH_POP()                                                                       \
  do {                                                                                       \
    pending = false;                                                                         \
    if (sp > 0) {                                                                            \
      const uint2 e_ = stack[(sp - 1) * ODW_MESH_THREADS];                                   \
      uint32_t sl_;                                                                          \
      ODW_MESH_FIRST(e_.y & 0xffu, sl_);                                                     \
      const uint32_t rest_ = (e_.y & 0xffu) & ~(1u << sl_);                                  \
      if (rest_) stack[(sp - 1) * ODW_MESH_THREADS] = make_uint2(e_.x, (e_.y & ~0xffu) | rest_); \
      else --sp;                                                                             \
      cur = (int)(e_.x + (uint32_t)__popc((e_.y >> 8) & ((1u << sl_) - 1u)));                \
      walking = true;                                                                        \
    } else {                                                                                 \
      cur = -1;                                                                              \
      walking = false;                                                                       \
    }                                                                                        \
  } while (0)

#ifdef ODW_MESH_STATS
  uint64_t phase_t[6] = {0, 0, 0, 0, 0, 0};
  uint64_t t_mark = __builtin_readcyclecounter();
#endif
  for (;;) {
    // ---- A: new rays for idle lanes, from the wave's ring ------------------------------------------
    const uint64_t idle = __ballot(!alive);
    if (idle == ~0ull && drained && ring_n == 0) break;    // nothing live, nothing left
    if (idle && !(drained && ring_n == 0)) {
      if (ring_n == 0) {
        if (next >= chunk_end) {
          unsigned long long c = 0;
          if (lane == 0) c = atomicAdd(P.out.chunk_counter, 1ull);
          const uint64_t chunk = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)c);
          next = chunk * (uint64_t)P.chunk;
          if (next >= P.n_rays) { next = P.n_rays; drained = true; }
          chunk_end = next + P.chunk < P.n_rays ? next + P.chunk : P.n_rays;
        }
        const uint64_t avail = chunk_end - next;
        const uint32_t fill = avail < ODW_MESH_RING ? (uint32_t)avail : (uint32_t)ODW_MESH_RING;
        ODW_MSTAT(0, __ballot(lane < fill));
        if (lane < fill) {
          // (position next + lane of the hand-out order is ray number r: the order groups rays that start alike)
          const uint64_t r = P.ray_order ? (uint64_t)P.ray_order[next + lane] : next + lane;
          d3 o, d;
          if (P.ray_origins) {
            o = mk(P.ray_origins[r], P.ray_origins[P.ray_stride + r], P.ray_origins[2 * P.ray_stride + r]);
            d = mk(P.ray_dirs[r], P.ray_dirs[P.ray_stride + r], P.ray_dirs[2 * P.ray_stride + r]);
            d = d * (1.0 / sqrt(dot(d, d)));
          } else {
            const RayInit g = generate_ray(P.source, P.first_ray + r, P.seed);
            o = g.point; d = g.dir;
          }
          double* slot = ring + 7 * lane;
          slot[0] = o.x; slot[1] = o.y; slot[2] = o.z; slot[3] = d.x; slot[4] = d.y; slot[5] = d.z;
          slot[6] = __longlong_as_double((long long)r);
        }
        __builtin_amdgcn_wave_barrier();
        ring_base = next;
        ring_n = fill;
        next += fill;
      }
      const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
      const uint32_t want = __popcll(idle);
      const uint32_t take = want < ring_n ? want : ring_n;
      if (!alive && rank < take) {
        const uint32_t s = ring_n - 1 - rank;
        const double* slot = ring + 7 * s;
        point = mk(slot[0], slot[1], slot[2]);
        dir = mk(slot[3], slot[4], slot[5]);
        i = (uint64_t)__double_as_longlong(slot[6]);
        power = P.ray_origins ? (P.ray_powers ? P.ray_powers[i] : 1.0) : as_const(P.source)->power;
        seq = 0; nint = 0; medium = -1; skip = -1; inside = -1;
        alive = true; fresh = true; walking = false; pending = false;
      }
      ring_n -= take;
    }
    ODW_MTIME(0);
    // ---- B: set up the next segment of fresh lanes ---------------------------------------------
    ODW_MSTAT(1, __ballot(alive && fresh));
    if (alive && fresh) {
      fresh = false;
      if (nint >= lim.max_intersections) {
        ODW_MCOUNT(ODW_CNT_CAPPED);
        atomicAdd(&wave_cnt[ODW_CNT_SEGMENTS], (uint32_t)nint);
        ODW_MCOUNT(ODW_CNT_TRACED_RAYS);
        alive = false;
      } else {
        ++nint;
        mask = sc.all_mask;
        if (sc.seq_enabled) mask = (seq < sc.seq_len) ? seq_mask[seq] : 0ull;
        mask &= ~sc.ignore_mask;
        q.any.t = INFINITY; q.any.prim = 0x7fffffff; q.any.face = 0x7fffffff;
        q.oth = q.any;
        walking = false; pending = false;
        if (mask != 0ull) {
          // where the ray enters the tree's box (float64): the float32 walk starts there, with the origin
          // as near to the geometry as it gets; a ray that misses the box has no candidates
          const d3 inv = mk(frcp(dir.x), frcp(dir.y), frcp(dir.z));
          double t0 = 0.0, t1 = q.tmax;
          {
            const double a0 = (sc.wide_lo[0] - point.x) * inv.x, a1 = (sc.wide_hi[0] - point.x) * inv.x;
            const double b0 = (sc.wide_lo[1] - point.y) * inv.y, b1 = (sc.wide_hi[1] - point.y) * inv.y;
            const double c0 = (sc.wide_lo[2] - point.z) * inv.z, c1 = (sc.wide_hi[2] - point.z) * inv.z;
            t0 = fmax(t0, fmax(fmin(a0, a1), fmax(fmin(b0, b1), fmin(c0, c1))));
            t1 = fmin(t1, fmin(fmax(a0, a1), fmin(fmax(b0, b1), fmax(c0, c1))));
          }
          if (t0 <= t1 * (1.0 + 1e-9) + 1e-6) {
            // (rounded down: the moved origin is a point of the ray before the box, exactly)
            tsh = t0 > 2e-3 ? (float)(t0 - 1e-3) * 0.999999f : 0.0f;
            const d3 o = point + dir * (double)tsh;
            ofx = (float)o.x; ofy = (float)o.y; ofz = (float)o.z;
            ivx = (float)inv.x; ivy = (float)inv.y; ivz = (float)inv.z;
            dfx = (float)dir.x; dfy = (float)dir.y; dfz = (float)dir.z;
            // (the direction in units of 1 / 127, one signed byte per axis, 127 in the fourth: the cone test is one v_dot4)
            dq = (int)(((uint32_t)(int)rintf(127.0f * dfx) & 0xffu) | (((uint32_t)(int)rintf(127.0f * dfy) & 0xffu) << 8) |
                       (((uint32_t)(int)rintf(127.0f * dfz) & 0xffu) << 16) | (127u << 24));
            oct = (dir.x > 0 ? 1u : 0u) | (dir.y > 0 ? 2u : 0u) | (dir.z > 0 ? 4u : 0u);
            cutf = (float)(q.tmax - (double)tsh) * 1.00001f + 1e-3f;
            cur = 0; sp = 0;
            walking = true;
          }
        }
      }
    }
    ODW_MTIME(1);
    // ---- C: node steps, all walking lanes together -------------------------------------------------
    for (int it = 0;; ++it) {
      const uint64_t wb = __ballot(walking);
      if (wb == 0ull) break;
      if (it > 0 && __popcll(wb) < ODW_MESH_STEP_MIN &&
          (__ballot(alive && !walking) != 0ull || (!(drained && ring_n == 0) && __ballot(!alive) != 0ull)))
        break;
      ODW_MSTAT(2, wb);
      if (walking) {
        cu32 nd = nodes + (size_t)cur * ODW_WIDE_WORDS;
        const vu4 h0 = *reinterpret_cast<const vu4 ODW_CONST*>(nd);          // corner, exponents
        const vu4 h1 = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 4);      // first inner child, first leaf record, slots, counts
        const vu4 qa = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 8);      // near corner: x x y y
        const vu4 qb = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 12);     // near z z | far x x
        const vu4 qc = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 16);     // far y y z z
        // t of plane q on axis a: q * (scale_a / d_a) + (corner_a - o_a) / d_a
        const float ax = __uint_as_float((h0.w & 0xffu) << 23) * ivx, bx = (__uint_as_float(h0.x) - ofx) * ivx;
        const float ay = __uint_as_float(((h0.w >> 8) & 0xffu) << 23) * ivy, by = (__uint_as_float(h0.y) - ofy) * ivy;
        const float az = __uint_as_float(((h0.w >> 16) & 0xffu) << 23) * ivz, bz = (__uint_as_float(h0.z) - ofz) * ivz;
        // per axis: the planes the ray meets first / last
        const uint32_t nx0 = ivx < 0 ? qb.z : qa.x, nx1 = ivx < 0 ? qb.w : qa.y, fx0 = ivx < 0 ? qa.x : qb.z, fx1 = ivx < 0 ? qa.y : qb.w;
        const uint32_t ny0 = ivy < 0 ? qc.x : qa.z, ny1 = ivy < 0 ? qc.y : qa.w, fy0 = ivy < 0 ? qa.z : qc.x, fy1 = ivy < 0 ? qa.w : qc.y;
        const uint32_t nz0 = ivz < 0 ? qc.z : qb.x, nz1 = ivz < 0 ? qc.w : qb.y, fz0 = ivz < 0 ? qb.x : qc.z, fz1 = ivz < 0 ? qb.y : qc.w;
        uint32_t hits = 0;
#define ODW_MESH_SLOT(S, NX, NY, NZ, FX, FY, FZ, SH)                                                          \
        {                                                                                                     \
          const float tnx = fmaf((float)(((NX) >> (SH)) & 0xffu), ax, bx), tfx = fmaf((float)(((FX) >> (SH)) & 0xffu), ax, bx); \
          const float tny = fmaf((float)(((NY) >> (SH)) & 0xffu), ay, by), tfy = fmaf((float)(((FY) >> (SH)) & 0xffu), ay, by); \
          const float tnz = fmaf((float)(((NZ) >> (SH)) & 0xffu), az, bz), tfz = fmaf((float)(((FZ) >> (SH)) & 0xffu), az, bz); \
          const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);                     \
          /* conservative acceptance: relative 1e-5 + absolute 1e-3 mm on t (as in nearest<true>) */         \
          const bool h_ = tf * 1.00001f + 1e-3f >= fmaxf(tn, 0.f) * 0.99999f - 1e-3f && tn * 0.99999f - 1e-3f < cutf; \
          hits |= h_ ? (1u << (S)) : 0u;                                                                      \
        }
        ODW_MESH_SLOT(0, nx0, ny0, nz0, fx0, fy0, fz0, 0)
        ODW_MESH_SLOT(1, nx0, ny0, nz0, fx0, fy0, fz0, 8)
        ODW_MESH_SLOT(2, nx0, ny0, nz0, fx0, fy0, fz0, 16)
        ODW_MESH_SLOT(3, nx0, ny0, nz0, fx0, fy0, fz0, 24)
        ODW_MESH_SLOT(4, nx1, ny1, nz1, fx1, fy1, fz1, 0)
        ODW_MESH_SLOT(5, nx1, ny1, nz1, fx1, fy1, fz1, 8)
        ODW_MESH_SLOT(6, nx1, ny1, nz1, fx1, fy1, fz1, 16)
        ODW_MESH_SLOT(7, nx1, ny1, nz1, fx1, fy1, fz1, 24)
#undef ODW_MESH_SLOT
        if (skip >= 0) {
          // the convex solid the ray has just left: children whose primitives all belong to it are not looked at
          const vu4 so = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 20);
          const uint32_t sk = (uint32_t)skip;
          hits &= ~(((so.x & 0xffffu) == sk ? 1u : 0u) | ((so.x >> 16) == sk ? 2u : 0u) | ((so.y & 0xffffu) == sk ? 4u : 0u) |
                    ((so.y >> 16) == sk ? 8u : 0u) | ((so.z & 0xffffu) == sk ? 16u : 0u) | ((so.z >> 16) == sk ? 32u : 0u) |
                    ((so.w & 0xffffu) == sk ? 64u : 0u) | ((so.w >> 16) == sk ? 128u : 0u));
        }
#if ODW_MESH_CONES
        if (inside >= 0 && hits) {
          // inside a convex solid: slots whose facets all belong to it and all face the ray (they could only be met from
          // outside) are dropped -- the neighbourhood of the facet the segment starts on, for one
          const vu4 so = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 20);
          const vu4 ca = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 24);
          const vu4 cb = *reinterpret_cast<const vu4 ODW_CONST*>(nd + 28);
          const uint32_t in = (uint32_t)inside;
#define ODW_MESH_CONE(S, W, SOLID)                                                                                      \
          {                                                                                                             \
            const bool drop = (SOLID) == in && __builtin_amdgcn_sdot4((int)(W), dq, 0, false) < 0;                       \
            hits &= drop ? ~(1u << (S)) : ~0u;                                                                          \
          }
          ODW_MESH_CONE(0, ca.x, so.x & 0xffffu)
          ODW_MESH_CONE(1, ca.y, so.x >> 16)
          ODW_MESH_CONE(2, ca.z, so.y & 0xffffu)
          ODW_MESH_CONE(3, ca.w, so.y >> 16)
          ODW_MESH_CONE(4, cb.x, so.z & 0xffffu)
          ODW_MESH_CONE(5, cb.y, so.z >> 16)
          ODW_MESH_CONE(6, cb.z, so.w & 0xffffu)
          ODW_MESH_CONE(7, cb.w, so.w >> 16)
#undef ODW_MESH_CONE
        }
#endif
        const uint32_t imask = h1.z & 0xffu, lmask = (h1.z >> 8) & 0xffu;
        const uint32_t ih = hits & imask, lh = hits & lmask;
        if (lh) {
          // leaves first (their hits cull the inner children): the lane waits for phase D
          lbase = h1.y; lcounts = h1.w; lhits = lh;
          if (ih) { stack[sp * ODW_MESH_THREADS] = make_uint2(h1.x, (imask << 8) | ih); ++sp; }
          walking = false;
          pending = true;
        } else if (ih) {
          uint32_t sl;
          ODW_MESH_FIRST(ih, sl);
          const uint32_t rest = ih & ~(1u << sl);
          if (rest) { stack[sp * ODW_MESH_THREADS] = make_uint2(h1.x, (imask << 8) | rest); ++sp; }
          cur = (int)(h1.x + (uint32_t)__popc(imask & ((1u << sl) - 1u)));
        } else {
          ODW_MESH_POP();
        }
      }
    }
    ODW_MTIME(2);
    // ---- D: leaves, then the interaction of the lanes whose traversal is over ------------------------
    ODW_MSTAT(3, __ballot(alive && pending));
    if (alive && pending) {
      // the ray, moved to the point next to the centre of this node's leaves and taken relative to it (float64, then
      // rounded: magnitudes of the size of the group)
      const vf4 hq = *reinterpret_cast<const vf4 ODW_CONST*>(leaves + (size_t)lbase * ODW_LEAF_WORDS + 12);
      const d3 oc = point - mk((double)hq.y, (double)hq.z, (double)hq.w);
      const double tc = -dot(oc, dir);
      const d3 orel = oc + dir * tc;
      const float ox = (float)orel.x, oy = (float)orel.y, oz = (float)orel.z;
      const float dx = dfx, dy = dfy, dz = dfz;
      // the records of the leaf children that were hit, as bits relative to the node's first record (<= 64 per node):
      // a leaf's records follow those of the leaf slots below it (sum of their 4-bit counts)
      uint64_t cand = 0;
      if (have_cand) {
        cand = ((uint64_t)lhits << 32) | lcounts;            // (what an earlier round left over)
      } else {
        for (uint32_t todo = lhits; todo; todo &= todo - 1u) {
          const uint32_t sl = (uint32_t)__ffs((int)todo) - 1u;
          const uint32_t cnt = (lcounts >> (4u * sl)) & 15u;
          const uint32_t below = lcounts & ((1u << (4u * sl)) - 1u);
          const uint32_t pairs = (below & 0x0f0f0f0fu) + ((below >> 4) & 0x0f0f0f0fu);
          cand |= ((1ull << cnt) - 1ull) << ((pairs * 0x01010101u) >> 24);
        }
      }
      ODW_MSUM(6, __popcll(cand));
      cf32 rec0 = leaves + (size_t)lbase * ODW_LEAF_WORDS;
      const auto filter = [&](vf4 w0, vf4 w1, vf4 w2) -> bool {
        // w0: v0x v0y v0z e1x, w1: e1y e1z e2x e2y, w2: e2z gs smax err.  Geometry only: whether the facet's group is
        // relevant and its solid not the one just left is asked of the few that pass
        if (__float_as_uint(w2.y) >> 31) return true;
        const float e1x = w0.w, e1y = w1.x, e1z = w1.y, e2x = w1.z, e2y = w1.w, e2z = w2.x;
        const float pvx = dy * e2z - dz * e2y, pvy = dz * e2x - dx * e2z, pvz = dx * e2y - dy * e2x;
        const float det = e1x * pvx + e1y * pvy + e1z * pvz;
        const float tvx = ox - w0.x, tvy = oy - w0.y, tvz = oz - w0.z;
        const float qvx = tvy * e1z - tvz * e1y, qvy = tvz * e1x - tvx * e1z, qvz = tvx * e1y - tvy * e1x;
        float U = tvx * pvx + tvy * pvy + tvz * pvz;
        float V = dx * qvx + dy * qvy + dz * qvz;
        const float ad = fabsf(det);
        U = det < 0 ? -U : U;
        V = det < 0 ? -V : V;
        // bounds: barycentric slack of the tolerance rules (intersect_tri) + rounding of this arithmetic
        const float m = tolf * w2.z + 1e-6f;
        const float ea = (fabsf(tvx) + fabsf(tvy) + fabsf(tvz)) * w2.w + m * ad;
        // (tried: the distance as a third condition, t in (tol, nearest + 2 tol) -- 8 % slower: few facets fail on it
        //  alone, all pay for it)
        return U >= -ea && V >= -ea && U + V <= ad + 2.0f * ea;
      };
      uint64_t pass = 0;
#if ODW_MESH_CAND_TRIPS > 0
      // at most ODW_MESH_CAND_TRIPS trips per round: a lane with many candidates goes on in the next round and does
      // not hold the lanes with few (they walk or interact meanwhile)
      for (int trip = 0; cand && trip < ODW_MESH_CAND_TRIPS; ++trip) {
#else
      while (cand) {
#endif
        // four records per trip: twelve loads in flight before the first is used (the trip is a round trip to the
        // cache: fewer, fuller trips)
        int kk[4];
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          on[j] = cand != 0ull;
          kk[j] = on[j] ? __ffsll((unsigned long long)cand) - 1 : kk[0];
          cand &= cand - 1ull;                                // (0 stays 0)
        }
        vf4 w[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cf32 r = rec0 + (size_t)kk[j] * ODW_LEAF_WORDS;
          w[j][0] = *reinterpret_cast<const vf4 ODW_CONST*>(r);
          w[j][1] = *reinterpret_cast<const vf4 ODW_CONST*>(r + 4);
          w[j][2] = *reinterpret_cast<const vf4 ODW_CONST*>(r + 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pass |= (on[j] && filter(w[j][0], w[j][1], w[j][2])) ? (1ull << kk[j]) : 0ull;
      }
      ODW_MSTAT(4, __ballot(pass != 0ull));
      ODW_MSUM(7, __popcll(pass));
      while (pass) {
        const int k = __ffsll((unsigned long long)pass) - 1;
        pass &= pass - 1ull;
        cf32 rec = rec0 + (size_t)k * ODW_LEAF_WORDS;
        const int p = (int)__float_as_uint(rec[12]);
        const uint32_t gs = __float_as_uint(rec[9]);
        if (!(((mask >> (gs & 0xff)) & 1) && (int)((gs >> 8) & 0x7fff) != skip)) continue;
        if (gs >> 31) {
          const MeshBest b = mesh_intersect_prim((ckargs)__builtin_amdgcn_kernarg_segment_ptr(), q.start, q.dn, q.tol, q.tmax, q.medium,
                                                 q.any, q.oth, p);
          q.any = b.any; q.oth = b.oth;
        } else {
          intersect_tri(sv, q, p, (int)(gs & 0xff));
        }
      }
      cutf = (float)(fmin(q.tmax, q.any.t + 2.0 * q.tol) - (double)tsh) * 1.00001f + 1e-3f;
      if (cand) {
        lcounts = (uint32_t)cand; lhits = (uint32_t)(cand >> 32);
        have_cand = true;                                      // (still pending)
      } else {
        have_cand = false;
        ODW_MESH_POP();
      }
    }
    ODW_MTIME(3);
    ODW_MSTAT(5, __ballot(alive && !walking && !pending && !fresh));
    // the interaction waits until ODW_MESH_INTERACT_MIN lanes are done with the tree, unless nobody walks or tests
    // leaves any more (1: at once)
    // Rays handed out in sorted order (ray_order): the lanes of a wave hold neighbouring rays -- interacting TOGETHER
    // keeps them in step for the next segment too (the same nodes, the same leaves, one wave-wide packet).  Measured
    // (1e7 rays, 6.5e4 / 1e6 facets, ms): at once 7.9 / 11.8, when 32 lanes are done 6.8 / 10.3, 60: 6.3 / 9.9, all: 5.0 / 7.4.
    // Rays in index order (explicit rays, surface sources) are unrelated: each lane goes on as soon as it is done.
    const uint64_t done_b = __ballot(alive && !walking && !pending && !fresh);
    const bool go = interact_min <= 1u || (uint32_t)__popcll(done_b) >= interact_min || __ballot(walking || pending) == 0ull;
    if (go && alive && !walking && !pending && !fresh) {
      if (q.any.prim == 0x7fffffff) {
        ODW_MCOUNT(ODW_CNT_ESCAPED);
        alive = false;
      } else {
        const bool use_oth = q.oth.prim != 0x7fffffff && q.oth.t < q.any.t + 2.0 * q.tol;
        const MeshRay r = mesh_interact<STOCH>((ckargs)__builtin_amdgcn_kernarg_segment_ptr(), point, dir, power, medium, seq, nint, i,
                                               use_oth ? q.oth.t : q.any.t, use_oth ? q.oth.prim : q.any.prim,
                                               use_oth ? q.oth.face : q.any.face, wave_cnt, hit_state, group_f64, group_i32, group_gdir);
        point = r.point; dir = r.dir; power = r.power; medium = r.medium; seq = r.seq; skip = r.skip; inside = r.inside; alive = r.alive;
        fresh = alive;
      }
      if (!alive) {
        atomicAdd(&wave_cnt[ODW_CNT_SEGMENTS], (uint32_t)nint);
        ODW_MCOUNT(ODW_CNT_TRACED_RAYS);
      }
    }
    ODW_MTIME(5);
  }
#undef ODW_MESH_POP
#undef ODW_MESH_FIRST
  // slots of the last block this wave never filled (as in odw_trace_kernel)
  const uint32_t hit_used = hit_state[2];
  const uint64_t hit_base = ((uint64_t)hit_state[1] << 32) | hit_state[0];
  if (P.out.hit_block && hit_used < P.out.hit_block) {
    const uint32_t left = P.out.hit_block - hit_used;
    for (uint32_t k = __lane_id(); k < left; k += 64)
      if (hit_base + hit_used + k < P.out.hit_capacity) P.out.hits[hit_base + hit_used + k].tag = ODW_TAG_UNUSED;
    const uint64_t at = hit_base + hit_used;
    const uint64_t in_buf = at < P.out.hit_capacity ? (P.out.hit_capacity - at < left ? P.out.hit_capacity - at : left) : 0;
    if (__lane_id() == 0 && in_buf) atomicAdd(P.out.hit_count + 1, (unsigned long long)in_buf);
  }
#ifdef ODW_MESH_STATS
  if (lane < 16 && P.dbg) atomicAdd(P.dbg + lane, (unsigned long long)wave_cnt[8 + lane]);
  if (lane == 0 && P.dbg)
    for (int k = 0; k < 6; ++k) atomicAdd(P.dbg + 16 + k, (unsigned long long)phase_t[k]);
#endif
  // (one atomic per block and counter: the waves' words are next to each other in LDS)
  __syncthreads();
  if (threadIdx.x < ODW_CNT_LDS) {
    uint32_t s = 0;
    for (int w = 0; w < ODW_MESH_BLOCK_WAVES; ++w) s += lds32[word_off + w * ODW_MESH_WAVE_WORDS + threadIdx.x];
    if (s) atomicAdd(P.out.counters + threadIdx.x, (unsigned long long)s);
  }
}


// ---- the order rays are handed out in (TraceParams.ray_order) ---------------------------------------------------------
// key of ray r: where it starts (10 bits per axis of the tree's root box, clamped; Morton order) above where it points
// (octahedral map of the unit direction, 16 bits per axis, Morton order).  Sorted by it, the 64 rays of a wave are
// neighbours among ALL rays of the launch: their node and leaf fetches fall into the same cache lines and their
// traversals take the same turns.
__device__ __forceinline__ uint32_t spread16(uint32_t v) {        // abcd -> 0a0b0c0d
  v &= 0xffffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__device__ __forceinline__ uint32_t spread10(uint32_t v) {        // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
// K = uint32_t: the direction bits alone (a point source at its focus: every ray starts at one point)
template <class K>
__global__ __launch_bounds__(256) void odw_ray_key_kernel(const DeviceSource* sp, uint64_t first, uint64_t n, uint64_t seed,
                                                          double lx, double ly, double lz, double sx, double sy, double sz,
                                                          K* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const RayInit g = generate_ray(sp, first + r, seed);
  const d3 d = g.dir;
  const double a = frcp(fabs(d.x) + fabs(d.y) + fabs(d.z));
  double u = d.x * a, v = d.y * a;
  if (d.z < 0) {
    const double fu = (1.0 - fabs(v)) * (u < 0 ? -1.0 : 1.0), fv = (1.0 - fabs(u)) * (v < 0 ? -1.0 : 1.0);
    u = fu; v = fv;
  }
  const uint32_t qu = (uint32_t)fmin(fmax((u * 0.5 + 0.5) * 65535.0, 0.0), 65535.0);
  const uint32_t qv = (uint32_t)fmin(fmax((v * 0.5 + 0.5) * 65535.0, 0.0), 65535.0);
  const uint32_t kd = spread16(qu) | (spread16(qv) << 1);
  if constexpr (sizeof(K) == 4) {
    keys[r] = kd;
  } else {
    const uint32_t qx = (uint32_t)fmin(fmax((g.point.x - lx) * sx, 0.0), 1023.0);
    const uint32_t qy = (uint32_t)fmin(fmax((g.point.y - ly) * sy, 0.0), 1023.0);
    const uint32_t qz = (uint32_t)fmin(fmax((g.point.z - lz) * sz, 0.0), 1023.0);
    const uint32_t ko = spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2);
    keys[r] = ((uint64_t)ko << 32) | kd;
  }
  vals[r] = (uint32_t)r;
}

}  // namespace odw
