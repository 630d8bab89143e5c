// odw_grid.hip -- trace kernel for big analytic scenes (hundreds of primitives, no facets):
// a rectilinear grid walked cell by cell (3-D DDA) instead of a BVH, and a per-lane state machine
// whose loop iteration is ONE CELL, not one segment.
//
// Why (hugeArray, 1500 spheres, profiles/r01): a BVH traversal costs each lane a different number
// of node visits, the lanes of a wave wait for the slowest (27 % of the issued lanes did work),
// and its per-thread node stack (32 KB of LDS per block) capped the occupancy.  A grid walk needs
// no stack; its state is a cell index and three plane distances.  With that little state the ray
// loop can be turned inside out: every iteration, every lane that is walking takes one cell step
// (uniform work: fetch the cell, test its primitives, advance); lanes whose walk has ended wait
// until enough of them have gathered, then the (divergent, expensive) interaction code runs for
// all of them at once and they start their next segment or take a new ray.
//
// Same rules as nearest<>() of odw_kernels.hip (ray.py:290-452): every candidate goes through
// consider(); the walk goes on while cells begin before nearest + 2 distTol.  Cells are visited in
// the order the ray meets them, a primitive is listed in every cell its box (tolerance slack
// included) touches, so every admissible candidate has been seen when the walk stops.
//
// Block = 1024 threads (16 waves, one block per CU, <= 128 VGPRs): bounds, cell table and --
// for sphere scenes -- the primitive records live in LDS (hugeArray: 0.2 + 6 + 72 KB).
#include "odw_device.h"

namespace odw {

#ifndef ODW_GRID_STEP_MIN
#define ODW_GRID_STEP_MIN 40     // keep stepping while at least this many lanes of the wave walk
#endif
#ifndef ODW_GRID_STEP_MAX
#define ODW_GRID_STEP_MAX 6      // ... but look at the waiting lanes at least every so many steps
#endif
#define ODW_GRID_THREADS 1024
#define ODW_GRID_WAVES (ODW_GRID_THREADS / 64)

struct GridView {
  const double* bx; const double* by; const double* bz;   // LDS
  const uint32_t* cells;
  const void* items;
  int nx, ny, nz;
};

// index i with b[i] <= v < b[i+1], clamped to [0, n-1]   (b has n+1 entries)
__device__ __forceinline__ int grid_slab(const double* b, int n, double v) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (v >= b[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

struct Walk {
  double tx, ty, tz;   // ray parameter at which the walk leaves the cell through each axis
  d3 inv;
  int cell;            // ix | iy << 8 | iz << 16
};

// start of a segment's walk: clip the ray to the grid, find the first cell.  false: misses the grid
__device__ __forceinline__ bool walk_begin(const GridView& G, d3 o, d3 d, double tmax, Walk& w) {
  w.inv = mk(frcp(d.x), frcp(d.y), frcp(d.z));
  double t0 = 0.0, t1 = tmax;
#define ODW_CLIP(O, D, INV, B, N)                                            \
  if ((D) != 0) {                                                            \
    const double a_ = ((B)[0] - (O)) * (INV), b_ = ((B)[N] - (O)) * (INV);   \
    t0 = fmax(t0, fmin(a_, b_));                                             \
    t1 = fmin(t1, fmax(a_, b_));                                             \
  } else if ((O) < (B)[0] || (O) > (B)[N]) {                                 \
    return false;                                                            \
  }
  ODW_CLIP(o.x, d.x, w.inv.x, G.bx, G.nx)
  ODW_CLIP(o.y, d.y, w.inv.y, G.by, G.ny)
  ODW_CLIP(o.z, d.z, w.inv.z, G.bz, G.nz)
#undef ODW_CLIP
  if (!(t0 <= t1)) return false;
  const d3 p = o + d * t0;
  const int ix = grid_slab(G.bx, G.nx, p.x), iy = grid_slab(G.by, G.ny, p.y), iz = grid_slab(G.bz, G.nz, p.z);
  w.cell = ix | (iy << 8) | (iz << 16);
  w.tx = d.x > 0 ? (G.bx[ix + 1] - o.x) * w.inv.x : (d.x < 0 ? (G.bx[ix] - o.x) * w.inv.x : INFINITY);
  w.ty = d.y > 0 ? (G.by[iy + 1] - o.y) * w.inv.y : (d.y < 0 ? (G.by[iy] - o.y) * w.inv.y : INFINITY);
  w.tz = d.z > 0 ? (G.bz[iz + 1] - o.z) * w.inv.z : (d.z < 0 ? (G.bz[iz] - o.z) * w.inv.z : INFINITY);
  return true;
}

// leave the cell through the nearest plane.  false: left the grid
__device__ __forceinline__ bool walk_advance(const GridView& G, d3 o, d3 d, Walk& w) {
  int ix = w.cell & 0xff, iy = (w.cell >> 8) & 0xff, iz = w.cell >> 16;
  if (w.tx <= w.ty && w.tx <= w.tz) {
    ix += d.x > 0 ? 1 : -1;
    if (ix < 0 || ix >= G.nx) return false;
    w.tx = (G.bx[ix + (d.x > 0 ? 1 : 0)] - o.x) * w.inv.x;
  } else if (w.ty <= w.tz) {
    iy += d.y > 0 ? 1 : -1;
    if (iy < 0 || iy >= G.ny) return false;
    w.ty = (G.by[iy + (d.y > 0 ? 1 : 0)] - o.y) * w.inv.y;
  } else {
    iz += d.z > 0 ? 1 : -1;
    if (iz < 0 || iz >= G.nz) return false;
    w.tz = (G.bz[iz + (d.z > 0 ? 1 : 0)] - o.z) * w.inv.z;
  }
  w.cell = ix | (iy << 8) | (iz << 16);
  return true;
}

// per-wave event counters in LDS: the interaction code runs under divergent control flow, so the
// lanes add with ds_add_u32 (a handful of events per ray)
#define ODW_GCOUNT(k) atomicAdd(&wave_cnt[(k)], 1u)

template <bool SPHERES, bool IN_LDS>
__global__ __launch_bounds__(ODW_GRID_THREADS) void odw_grid_kernel(const TraceParams P) {
  extern __shared__ double grid_lds[];
  const DeviceScene& sc = P.scene;
  const DeviceLimits& lim = P.lim;
  const DeviceGrid& GD = P.grid;
  // ---- LDS image: bounds | cells | items | per-wave counters | per-wave hit-block state ----
  const int nb = GD.nx + GD.ny + GD.nz + 3;
  const int ncell = GD.nx * GD.ny * GD.nz;
  double* l_bounds = grid_lds;
  uint32_t* l_words = reinterpret_cast<uint32_t*>(l_bounds + nb);
  uint32_t* wave_cnt_all = l_words;                                   // [ODW_GRID_WAVES][16]
  uint32_t* hit_state_all = wave_cnt_all + ODW_GRID_WAVES * 16;       // [ODW_GRID_WAVES][4]
  uint32_t* l_cells = hit_state_all + ODW_GRID_WAVES * 4;
  // items start on a 16-byte boundary
  double* l_items = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(l_cells + (IN_LDS ? ncell : 0)) + 15) & ~(uintptr_t)15);
  for (int k = threadIdx.x; k < nb; k += ODW_GRID_THREADS) l_bounds[k] = GD.bounds[k];
  for (int k = threadIdx.x; k < ODW_GRID_WAVES * 16; k += ODW_GRID_THREADS) wave_cnt_all[k] = 0;
  if (threadIdx.x < ODW_GRID_WAVES * 4) hit_state_all[threadIdx.x] = (threadIdx.x & 3) == 2 ? P.out.hit_block : 0u;
  if (IN_LDS) {
    for (int k = threadIdx.x; k < ncell; k += ODW_GRID_THREADS) l_cells[k] = GD.cells[k];
    const int n_words = SPHERES ? GD.n_items * 6 : (GD.n_items + 1) / 2;       // doubles
    const double* src = reinterpret_cast<const double*>(GD.items);
    for (int k = threadIdx.x; k < n_words; k += ODW_GRID_THREADS) l_items[k] = src[k];
  }
  __syncthreads();
  GridView G;
  G.bx = l_bounds; G.by = l_bounds + GD.nx + 1; G.bz = l_bounds + GD.nx + GD.ny + 2;
  G.nx = GD.nx; G.ny = GD.ny; G.nz = GD.nz;
  G.cells = IN_LDS ? l_cells : GD.cells;
  G.items = IN_LDS ? (const void*)l_items : GD.items;
  const int wave = threadIdx.x >> 6;
  uint32_t* wave_cnt = wave_cnt_all + wave * 16;
  volatile uint32_t* hit_state = hit_state_all + wave * 4;

  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_hdr = as_const(sc.prim_hdr);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  cf64 group_f64 = as_const(sc.group_f64);
  ci32 group_i32 = as_const(sc.group_i32);
  cf64 group_gdir = as_const(sc.group_gdir);
  cu64 seq_mask = as_const(sc.seq_mask);

  const uint32_t lane = __lane_id();
  uint64_t next = 0, chunk_end = 0;                        // wave-uniform
  // lane states: !alive (no ray) | fresh (ray or segment to set up) | walking | waiting for the
  // interaction (alive && !fresh && !walking)
  bool alive = false, fresh = false, walking = false;
  uint64_t i = 0;
  // the ray's position, direction and medium live in the query record (one copy)
  Query q;
  q.tol = lim.dist_tol; q.tmax = lim.max_ray_length + lim.dist_tol;
  q.start = mk(0, 0, 0); q.dn = mk(0, 0, 1); q.medium = -1;
  q.any.t = INFINITY; q.any.prim = 0x7fffffff; q.any.face = 0x7fffffff;
  q.oth = q.any;
  d3& point = q.start;
  d3& dir = q.dn;
  int& medium = q.medium;
  double power = 0;
  int seq = 0, nint = 0, skip = -1;
  uint64_t mask = 0;
  bool drained = false;                                    // wave-uniform: the launch has no rays left to hand out
  Walk w;
  w.tx = w.ty = w.tz = INFINITY; w.inv = mk(0, 0, 0); w.cell = 0;

  for (;;) {
    // ---- A: new rays for idle lanes (as in odw_trace_kernel) -------------------------------
    const uint64_t idle = __ballot(!alive);
    if (idle == ~0ull && drained) break;                   // nothing live, nothing left
    if (!drained && idle && (idle == ~0ull || __popcll(idle) >= ODW_REFILL_MIN)) {
      if (next >= chunk_end) {
        unsigned long long c = 0;
        if (lane == (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1)) c = atomicAdd(P.out.chunk_counter, 1ull);
        const uint64_t chunk = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                               __builtin_amdgcn_readfirstlane((uint32_t)c);
        next = chunk * ODW_CHUNK;
        if (next >= P.n_rays) { next = P.n_rays; drained = true; }
        chunk_end = next + ODW_CHUNK < P.n_rays ? next + ODW_CHUNK : P.n_rays;
      }
      const uint64_t avail = next < chunk_end ? chunk_end - next : 0;
      const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
      const uint32_t want = __popcll(idle);
      const uint32_t take = want < avail ? want : (uint32_t)avail;
      if (!alive && rank < take) {
        i = next + rank;
        if (P.ray_origins) {
          point = mk(P.ray_origins[3 * i], P.ray_origins[3 * i + 1], P.ray_origins[3 * i + 2]);
          dir = mk(P.ray_dirs[3 * i], P.ray_dirs[3 * i + 1], P.ray_dirs[3 * i + 2]);
          dir = dir * (1.0 / sqrt(dot(dir, dir)));
          power = P.ray_powers ? P.ray_powers[i] : 1.0;
        } else {
          const RayInit r = generate_ray(P.source, P.first_ray + i, P.seed);
          point = r.point; dir = r.dir; power = r.power;
        }
        seq = 0; nint = 0; medium = -1; skip = -1;
        alive = true; fresh = true; walking = false;
      }
      next += take;
      if (idle == ~0ull && drained) break;
    }
    // ---- B: set up the next segment of fresh lanes ---------------------------------------------
    if (alive && fresh) {
      fresh = false;
      if (nint >= lim.max_intersections) {
        ODW_GCOUNT(ODW_CNT_CAPPED);
        atomicAdd(&wave_cnt[ODW_CNT_SEGMENTS], (uint32_t)nint);
        ODW_GCOUNT(ODW_CNT_TRACED_RAYS);
        alive = false;
      } else {
        ++nint;
        mask = sc.all_mask;
        if (sc.seq_enabled) mask = (seq < sc.seq_len) ? seq_mask[seq] : 0ull;
        mask &= ~sc.ignore_mask;
        q.any.t = INFINITY; q.any.prim = 0x7fffffff; q.any.face = 0x7fffffff;
        q.oth = q.any;
        walking = mask != 0ull && walk_begin(G, point, dir, q.tmax, w);
      }
    }
    // ---- C: cell steps, all walking lanes together -----------------------------------------------
    for (int it = 0;; ++it) {
      const uint64_t wb = __ballot(walking);
      if (wb == 0ull) break;
      // stop stepping once few lanes walk (or after a few steps) if lanes wait: for their interaction,
      // or -- enough of them -- for a new ray
      if (it > 0 && (__popcll(wb) < ODW_GRID_STEP_MIN || it >= ODW_GRID_STEP_MAX) &&
          (__ballot(alive && !walking) != 0ull || (!drained && __popcll(__ballot(!alive)) >= ODW_REFILL_MIN)))
        break;
      if (walking) {
        const int ix = w.cell & 0xff, iy = (w.cell >> 8) & 0xff, iz = w.cell >> 16;
        const uint32_t word = G.cells[ix + G.nx * (iy + G.ny * iz)];
        const uint32_t first = word & 0xffffffu, count = word >> 24;
        for (uint32_t k = 0; k < count; ++k) {
          if (SPHERES) {
            const double2* rec = reinterpret_cast<const double2*>(reinterpret_cast<const double*>(G.items) + 6 * (size_t)(first + k));
            const double2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const uint64_t bits = (uint64_t)__double_as_longlong(r2.x);
            const int prim = (int)(uint32_t)bits, gs = (int)(uint32_t)(bits >> 32);
            const int g = gs & 0xff;
            if (((mask >> g) & 1) && (gs >> 8) != skip) {
              // a sphere needs no frame (as in intersect_prim): centre in global coordinates
              const d3 oc = q.start - mk(r0.x, r0.y, r1.x);
              double ta, tb;
              if (quad_roots_unit(dot(oc, q.dn), dot(oc, oc) - r1.y * r1.y, ta, tb) == 2) {
                const double bt = ta > q.tol ? ta : (tb > q.tol ? tb : INFINITY);
                consider(sv, q, bt, prim, 0, g, 0, 0);
              }
            }
          } else {
            const int p = (int)reinterpret_cast<const uint32_t*>(G.items)[first + k];
            ci32 pi = sv.prim_i32 + 4 * p;
            const int g = pi[1];
            if (((mask >> g) & 1) && (pi[2] >> ODW_SOLID_SHIFT) != skip) intersect_prim(sv, q, p, pi[0], g, pi[2], pi[3]);
          }
        }
        // candidates beyond nearest + 2 distTol can never be selected (ray.py:432, 440): the walk
        // ends when the cell it would enter next begins beyond that
        const double t_exit = fmin(w.tx, fmin(w.ty, w.tz));
        const double cut = fmin(q.tmax, q.any.t + 2.0 * q.tol);
        if (!(t_exit <= cut) || !walk_advance(G, q.start, q.dn, w)) walking = false;
      }
    }
    // ---- D: interaction of the lanes whose walk has ended -------------------------------------------
    if (alive && !walking && !fresh) {
      if (q.any.prim == 0x7fffffff) {
        ODW_GCOUNT(ODW_CNT_ESCAPED);
        alive = false;
      } else {
        const bool use_oth = q.oth.prim != 0x7fffffff && q.oth.t < q.any.t + 2.0 * q.tol;
        const double t_hit = use_oth ? q.oth.t : q.any.t;
        const int face = use_oth ? q.oth.face : q.any.face;
        const int prim = use_oth ? q.oth.prim : q.any.prim;
        cf64 pf = sv.prim_f64 + (size_t)prim * 16;
        ci32 pi = sv.prim_i32 + 4 * prim;
        point = point + dir * t_hit;
        if (medium >= 0) {                                  // ray.py:120-125 (assignment)
          const double L = group_f64[4 * medium + 2];
          if (L == 0) power = 0;
          else if (L < INFINITY) power = exp(-t_hit / L);
        }
        d3 n = face_normal(pi[0], pf + 12, face, xf_point(pf, point));
        if (pi[2] & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
        n = xf_vec_t(pf, n);
        const bool entering = dot(dir, n) < 0;
        if (entering) n = n * -1.0;
        const int g = pi[1];
        const int gtype = group_i32[4 * g];
        if (group_i32[4 * g + 1]) {
          ODW_GCOUNT(ODW_CNT_RECORDED_HITS);
          record_hit<true, 1, true>(P, P.first_ray + i, g, point, dir, power, entering, wave_cnt, hit_state);
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
              dir = line_grating(dir, nn, nn, n, P.wavelength, order, lpm, gd, false);
              ++seq;
            }
          } else if (entering) {
            if (medium >= 0) {
              atomicAdd(P.out.counters + ODW_CNT_GRATING_IN_MEDIUM, 1ull);     // a ValueError of the reference
              ODW_GCOUNT(ODW_CNT_DIED);
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
        skip = ((pi[2] & ODW_FLAG_CONVEX) && (entering ? -dot(dir, n) : dot(dir, n)) > 0) ? (pi[2] >> ODW_SOLID_SHIFT) : -1;
        if (alive && power < lim.power_tol) { ODW_GCOUNT(ODW_CNT_DIED); alive = false; }
        fresh = alive;
      }
      if (!alive) {
        atomicAdd(&wave_cnt[ODW_CNT_SEGMENTS], (uint32_t)nint);
        ODW_GCOUNT(ODW_CNT_TRACED_RAYS);
      }
    }
  }
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
  if (lane < ODW_CNT_LDS) {
    const uint32_t s = wave_cnt[lane];
    if (s) atomicAdd(P.out.counters + lane, (unsigned long long)s);
  }
}

}  // namespace odw
