// odw_grid.hip -- trace kernel for big analytic scenes (hundreds of primitives, no facets):
// a rectilinear grid walked cell by cell (3-D DDA) instead of a BVH, and a per-lane state machine
// whose loop iteration is ONE CELL, not one segment.
//
// Why (hugeArray, 1500 spheres, profiles/r01): a BVH traversal costs each lane a different number
// of node visits, the lanes of a wave wait for the slowest (27 % of the issued lanes did work),
// and its per-thread node stack (32 KB of LDS per block) capped the occupancy.  A grid walk needs
// no stack; its state is a cell index and three plane distances.  With that little state the ray
// loop can be turned inside out -- three queues inside every wave:
//   walk      every iteration, every walking lane takes one cell step: fetch the cell, a CHEAP test
//             of its primitives (sphere: sign of the discriminant; others: bounding box), advance.
//             Uniform work, all walking lanes together.
//   resolve   a lane whose cheap test says "maybe" stops walking.  When enough lanes wait, the
//   + interact  expensive, divergent code runs for all of them at once: exact roots and consider()
//             for the cell's primitives, then either back to walking or the interaction with the
//             surface (normal, Snell / mirror / grating, hit row) and the next segment's set-up.
//   generate  new rays come from a per-wave ring in LDS that the WHOLE wave fills, 64 rays at a
//             time (Philox, two table inversions, four sin/cos: ~600 instructions, now always at 64
//             lanes); an idle lane pops its next ray from the ring.
// Measured on hugeArray with one queue per segment (first version of this file): 1.79e10 VALU
// wave-instructions per 1.25e8 rays -- generation at 21 lanes per run, the root / consider path at
// ~9 lanes in nearly every cell step, interaction at 38.
//
// Same rules as nearest<>() of odw_kernels.hip (ray.py:290-452): every candidate goes through
// consider(); the walk goes on while cells begin before nearest + 2 distTol.  Cells are visited in
// the order the ray meets them, a primitive is listed in every cell its box (tolerance slack
// included) touches, so every admissible candidate has been seen when the walk stops.
//
// Block = 1024 threads (16 waves, one block per CU, <= 128 VGPRs): plane tables, cell table, -- for
// sphere scenes -- the primitive records and the ray rings live in LDS (hugeArray: 0.2 + 6 + 72 + 48 KB).
#include "odw_device.h"

namespace odw {

#ifndef ODW_GRID_STEP_MIN
#define ODW_GRID_STEP_MIN 24     // keep stepping while at least this many lanes of the wave walk (measured: 16-24 alike,
#endif                           // 32: +6 %, 40: +25 %, 56: +90 % time on hugeArray)
#ifndef ODW_GRID_REFILL_MIN
#define ODW_GRID_REFILL_MIN 1    // idle lanes that make the wave pop new rays from its ring (measured: 1: 25.8 ms,
                                 // 8: 39.8, 16: 41.5 per 1.25e8 hugeArray rays -- a pop costs ~30 instructions, a lane
                                 // that waits for seven others idles through whole cell steps)
#endif
#ifndef ODW_GRID_CUT_REG
#define ODW_GRID_CUT_REG 1       // where the walk ends (nearest + 2 distTol, the length limit) kept in a register pair, renewed when the nearest changes
#endif
#ifndef ODW_GRID_EXACT_WALK
#define ODW_GRID_EXACT_WALK 0    // A/B (round 4): exact sphere roots and consider() inside the cell step, no resolve queue
#endif
#ifndef ODW_GRID_LEAN_TEST
#define ODW_GRID_LEAN_TEST 1     // the walk's sphere test reads centre and radius only (A/B: 0 = the record's group / solid word too)
#endif
#ifndef ODW_GRID_SORTED
#define ODW_GRID_SORTED 0        // A/B (round 5, profiles/r05/README.md): 1 = also build the kernels that hand the launch's rays out sorted by
#endif                           // where they point (ODW_GRID_PRESORT = key bits) and let a wave's lanes interact together (ODW_GRID_GATE):
                                 // hugeArray 20.3 ms -> kernel 17.6 - 18.6 + key pass and radix sort 2.2 - 3.9 = 20.8 - 21.5 ms; not kept
#ifndef ODW_GRID_RCP
#define ODW_GRID_RCP frcp1          // (A/B: frcp = two Newton steps)
#endif
#define ODW_GRID_THREADS 1024
#define ODW_GRID_WAVES (ODW_GRID_THREADS / 64)
#define ODW_GRID_WAVE_WORDS 32   // per wave: event counters (0..7), diagnostics (8..27), hit-block state (28..31)
#define ODW_GRID_RING 64         // rays per wave in the ring (one per lane and fill)
#define ODW_GRID_RING_DOUBLES (ODW_GRID_RING * 7)   // per ray: origin, direction, first cell (found when the ring is filled)

// diagnostic build (-DODW_GRID_STATS): per phase, how often it ran and with how many lanes
#ifdef ODW_GRID_STATS
#define ODW_GSTAT(k, mask_)                                                                           \
  do {                                                                                                \
    const unsigned long long m_ = (mask_);                                                            \
    if (m_ && (int)__lane_id() == __ffsll(m_) - 1) { wave_cnt[8 + 2 * (k)] += 1u; wave_cnt[9 + 2 * (k)] += (uint32_t)__popcll(m_); } \
  } while (0)
// time per phase: s_memtime ticks between marks, kept per wave, added up at the end (odw_destroy prints)
#define ODW_GTIME(k) do { const uint64_t t_ = __builtin_readcyclecounter(); phase_t[(k)] += t_ - t_mark; t_mark = t_; } while (0)
#else
#define ODW_GSTAT(k, mask_) do {} while (0)
#define ODW_GTIME(k) do {} while (0)
#endif

// fmin as ONE v_min_f64.  The compiler puts a v_max_f64 x, x, x in front of every operand of fmin it cannot prove
// to be a quiet number (values that came through selects or LDS: the plane distances of the walk) -- five of the
// ~100 vector instructions of a cell step.  The instruction itself does what fmin does (IEEE mode: the operand that
// is a number wins).
__device__ __forceinline__ double fmin_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// index i with b[i] <= v < b[i+1], clamped to [0, n-1]   (b: n+1 planes in LDS)
__device__ __forceinline__ int grid_slab(const double* b, int n, double v) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (v >= b[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

// per-wave event counters in LDS: the interaction code runs under divergent control flow, so the
// lanes add with ds_add_u32 (a handful of events per ray)
#define ODW_GCOUNT(k) atomicAdd(&wave_cnt[(k)], 1u)

template <bool SPHERES, bool IN_LDS, bool SORTED>
__global__ __launch_bounds__(ODW_GRID_THREADS) void odw_grid_kernel(const TraceParams P) {
  extern __shared__ double grid_lds[];
  const DeviceScene& sc = P.scene;
  const DeviceLimits& lim = P.lim;
  const DeviceGrid& GD = P.grid;
  // ---- LDS image: planes | per-wave words | ray rings | cells | items ----
  // (addressed by offsets from the one dynamic array: pointers that pass through integers or
  //  through a struct lose their address space and turn every read into a flat load)
  const int nx = GD.nx, ny = GD.ny, nz = GD.nz;
  const int nb = nx + ny + nz + 3;
  const int ncell = nx * ny * nz;
  uint32_t* lds32 = reinterpret_cast<uint32_t*>(grid_lds);
  const int word_off = 2 * nb;                                            // [ODW_GRID_WAVES][ODW_GRID_WAVE_WORDS] words
  const int ring_off = (word_off + ODW_GRID_WAVES * ODW_GRID_WAVE_WORDS + 1) / 2;   // doubles: [ODW_GRID_WAVES][ring]
  const int cell_off = 2 * (ring_off + ODW_GRID_WAVES * ODW_GRID_RING_DOUBLES);      // words: [ncell]
  const int item_off = ((cell_off + (IN_LDS ? ncell : 0) + 3) & ~3) / 2;             // doubles, 16-byte aligned
  for (int k = threadIdx.x; k < nb; k += ODW_GRID_THREADS) grid_lds[k] = GD.bounds[k];
  for (int k = threadIdx.x; k < ODW_GRID_WAVES * ODW_GRID_WAVE_WORDS; k += ODW_GRID_THREADS)
    lds32[word_off + k] = (k % ODW_GRID_WAVE_WORDS) == 30 ? P.out.hit_block : 0u;   // hit-block state: base lo, hi, used (full: none reserved yet), -
  if (IN_LDS) {
    for (int k = threadIdx.x; k < ncell; k += ODW_GRID_THREADS) lds32[cell_off + k] = GD.cells[k];
    const int n_words = SPHERES ? GD.n_items * 6 : (GD.n_items + 1) / 2;       // doubles
    const double* src = reinterpret_cast<const double*>(GD.items);
    for (int k = threadIdx.x; k < n_words; k += ODW_GRID_THREADS) grid_lds[item_off + k] = src[k];
  }
  __syncthreads();
  const double* bx = grid_lds;                                   // the three plane tables, one after the other
  const int by_off = nx + 1, bz_off = nx + ny + 2;
  const int wave = threadIdx.x >> 6;
  uint32_t* wave_cnt = lds32 + word_off + wave * ODW_GRID_WAVE_WORDS;
  volatile uint32_t* hit_state = wave_cnt + 28;
  double* ring = grid_lds + ring_off + wave * ODW_GRID_RING_DOUBLES;      // slot s: origin (3), direction (3)

  SceneView sv;
  sv.prim_f64 = as_const(sc.prim_f64);
  sv.prim_hdr = as_const(sc.prim_hdr);
  sv.prim_i32 = as_const(sc.prim_i32);
  sv.cond_i32 = as_const(sc.cond_i32);
  // the groups' tables (64 x {ior, reflectivity, absorption length, lines per mm | optical type, record, grating type,
  // order | grating direction}: 4.5 KB) are read at every interaction with a per-lane index: from LDS, not through the
  // vector cache (a chain of two to three dependent global loads per interaction otherwise)
  __shared__ double group_f64[64 * 4];
  __shared__ int32_t group_i32[64 * 4];
  __shared__ double group_gdir[64 * 3];
  {
    cf64 gf = as_const(sc.group_f64);
    ci32 gi = as_const(sc.group_i32);
    cf64 gd = as_const(sc.group_gdir);
    for (int k = threadIdx.x; k < 64 * 4; k += ODW_GRID_THREADS) { group_f64[k] = gf[k]; group_i32[k] = gi[k]; }
    for (int k = threadIdx.x; k < 64 * 3; k += ODW_GRID_THREADS) group_gdir[k] = gd[k];
  }
  __syncthreads();
  cu64 seq_mask = as_const(sc.seq_mask);

  const uint32_t lane = __lane_id();
  uint64_t next = 0, chunk_end = 0;                        // wave-uniform: the wave's chunk of the launch
  uint64_t ring_base = 0;                                  // wave-uniform: ray index of ring slot 0
  uint32_t ring_n = 0;                                     // wave-uniform: rays left in the ring (slots 0 .. ring_n-1)
  bool drained = false;                                    // wave-uniform: the launch has no rays left to hand out
  // lane states: !alive (no ray) | fresh (segment to set up) | walking | pending (cheap test positive: resolve) |
  // waiting for the interaction (alive && !fresh && !walking && !pending)
  bool alive = false, fresh = false, walking = false, pending = false;
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
  int skip_rec = -1;      // SPHERES: the record of the sphere the ray has just left (skip is its solid), or -1
  uint64_t mask = 0;
  // the walk: ray parameter at which it leaves the cell through each axis, 1 / direction, the cell
  // (ix | iy << 8 | iz << 16)
  double tx = INFINITY, ty = INFINITY, tz = INFINITY, ivx = 0, ivy = 0, ivz = 0;
  double cut = 0;        // min(length limit, nearest + 2 distTol): cells that begin beyond it are not visited (ray.py:432, 440)
  int cell = 0;
  // (index order: the constants; a sorted launch reads its thresholds from the arguments -- A/B runs)
#define refill_min (SORTED ? P.refill_min : (uint32_t)ODW_GRID_REFILL_MIN)
#define interact_min (SORTED ? P.interact_min : 1u)

  // leave the cell through the nearest plane (the axis by selects, the next plane by ONE read of the
  // contiguous plane tables), or end the walk: beyond nearest + 2 distTol (ray.py:432, 440), or out of the grid
  // (tried in round 3: three predicated blocks, one per axis, each with its own operands instead of the selects --
  //  46 fewer v_cndmask in the binary, but three LDS reads in a row under divergent masks: 25.6 against 24.3 ms)
#define ODW_WALK_ADVANCE()                                                                   \
  do {                                                                                       \
    const double t_exit_ = fmin_raw(tx, fmin_raw(ty, tz));                                   \
    const double cut_ = ODW_GRID_CUT_REG ? cut : fmin_raw(q.tmax, q.any.t + 2.0 * q.tol);   \
    if (!(t_exit_ <= cut_)) {                                                                \
      walking = false;                                                                       \
    } else {                                                                                 \
      const bool ax_ = tx <= ty && tx <= tz;                                                 \
      const bool ay_ = !ax_ && ty <= tz;                                                     \
      const int shift_ = ax_ ? 0 : (ay_ ? 8 : 16);                                           \
      const int n_ = ax_ ? nx : (ay_ ? ny : nz);                                             \
      const double da_ = ax_ ? dir.x : (ay_ ? dir.y : dir.z);                                \
      const double oa_ = ax_ ? point.x : (ay_ ? point.y : point.z);                          \
      const double ia_ = ax_ ? ivx : (ay_ ? ivy : ivz);                                      \
      const int base_ = ax_ ? 0 : (ay_ ? by_off : bz_off);                                   \
      const int up_ = da_ > 0 ? 1 : 0;                                                       \
      const int idx_ = ((cell >> shift_) & 0xff) + 2 * up_ - 1;                              \
      if (idx_ < 0 || idx_ >= n_) {                                                          \
        walking = false;                                                                     \
      } else {                                                                               \
        const double t_ = (bx[base_ + idx_ + up_] - oa_) * ia_;                              \
        tx = ax_ ? t_ : tx;                                                                  \
        ty = ay_ ? t_ : ty;                                                                  \
        tz = (!ax_ && !ay_) ? t_ : tz;                                                       \
        cell = (cell & ~(0xff << shift_)) | (idx_ << shift_);                                \
        walking = true;                                                                      \
      }                                                                                      \
    }                                                                                        \
  } while (0)

#ifdef ODW_GRID_STATS
  uint64_t phase_t[6] = {0, 0, 0, 0, 0, 0};
  uint64_t t_mark = __builtin_readcyclecounter();
#endif
  for (;;) {
    // ---- A: new rays for idle lanes, from the wave's ring ------------------------------------------
    const uint64_t idle = __ballot(!alive);
    if (idle == ~0ull && drained && ring_n == 0) break;    // nothing live, nothing left
    if (idle && !(drained && ring_n == 0) && (idle == ~0ull || (uint32_t)__popcll(idle) >= refill_min)) {
      if (ring_n == 0) {
        // fill: the whole wave generates the next (up to) 64 rays of its chunk, one per lane
        if (next >= chunk_end) {
          // next chunk of the launch: one atomic per wave and chunk
          unsigned long long c = 0;
          if (lane == 0) c = atomicAdd(P.out.chunk_counter, 1ull);
          const uint64_t chunk = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)c);
          next = chunk * (uint64_t)P.chunk;
          if (next >= P.n_rays) { next = P.n_rays; drained = true; }
          chunk_end = next + P.chunk < P.n_rays ? next + P.chunk : P.n_rays;
        }
        const uint64_t avail = chunk_end - next;
        const uint32_t fill = avail < ODW_GRID_RING ? (uint32_t)avail : (uint32_t)ODW_GRID_RING;
        ODW_GSTAT(0, __ballot(lane < fill));
        if (lane < fill) {
          // (position next + lane of the hand-out order is ray number r: odw_capi.hip, presort_rays)
          const uint64_t r = SORTED ? (uint64_t)P.ray_order[next + lane] : next + lane;
          d3 o, d;
          if (P.ray_origins) {
            o = mk(P.ray_origins[r], P.ray_origins[P.ray_stride + r], P.ray_origins[2 * P.ray_stride + r]);
            d = mk(P.ray_dirs[r], P.ray_dirs[P.ray_stride + r], P.ray_dirs[2 * P.ray_stride + r]);
            d = d * (1.0 / sqrt(dot(d, d)));
          } else {
            const RayInit g = generate_ray(P.source, P.first_ray + r, P.seed);
            o = g.point; d = g.dir;
          }
          // the cell the ray starts in (clipped to the grid), found here, where all 64 lanes generate together,
          // and not at the set-up of the first segment, which shares its run with lanes that go on from a hit
          uint32_t first_cell = 0xffffffffu;
          {
            const double jx = ODW_GRID_RCP(d.x), jy = ODW_GRID_RCP(d.y), jz = ODW_GRID_RCP(d.z);
            bool in = true;
            double t0 = 0.0, t1 = q.tmax;
#define ODW_CLIP(O, D, INV, LO, HI)                                          \
            if ((D) != 0) {                                                  \
              const double a_ = ((LO) - (O)) * (INV), b_ = ((HI) - (O)) * (INV); \
              t0 = fmax(t0, fmin(a_, b_));                                   \
              t1 = fmin(t1, fmax(a_, b_));                                   \
            } else if ((O) < (LO) || (O) > (HI)) {                           \
              in = false;                                                    \
            }
            ODW_CLIP(o.x, d.x, jx, bx[0], bx[nx])
            ODW_CLIP(o.y, d.y, jy, bx[by_off], bx[by_off + ny])
            ODW_CLIP(o.z, d.z, jz, bx[bz_off], bx[bz_off + nz])
#undef ODW_CLIP
            if (in && t0 <= t1) {
              const d3 p0 = o + d * t0;
              first_cell = (uint32_t)grid_slab(bx, nx, p0.x) | ((uint32_t)grid_slab(bx + by_off, ny, p0.y) << 8) |
                           ((uint32_t)grid_slab(bx + bz_off, nz, p0.z) << 16);
            }
          }
          double* slot = ring + 7 * lane;
          slot[0] = o.x; slot[1] = o.y; slot[2] = o.z; slot[3] = d.x; slot[4] = d.y; slot[5] = d.z;
          reinterpret_cast<uint32_t*>(slot + 6)[0] = first_cell;
          if (SORTED) reinterpret_cast<uint32_t*>(slot + 6)[1] = (uint32_t)r;  // (a sorted launch holds < 2^31 rays)
        }
        // (one wave: its LDS operations complete in program order; this only keeps the compiler from
        //  moving the reads below above the writes)
        __builtin_amdgcn_wave_barrier();
        ring_base = next;
        ring_n = fill;
        next += fill;
      }
      // pop: idle lanes take the top rays of the ring
      const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
      const uint32_t want = __popcll(idle);
      const uint32_t take = want < ring_n ? want : ring_n;
      if (!alive && rank < take) {
        const uint32_t s = ring_n - 1 - rank;
        const double* slot = ring + 7 * s;
        point = mk(slot[0], slot[1], slot[2]);
        dir = mk(slot[3], slot[4], slot[5]);
        cell = (int)reinterpret_cast<const uint32_t*>(slot + 6)[0];       // (-1: the ray misses the grid)
        i = SORTED ? (uint64_t)reinterpret_cast<const uint32_t*>(slot + 6)[1] : ring_base + s;
        power = P.ray_origins ? (P.ray_powers ? P.ray_powers[i] : 1.0) : as_const(P.source)->power;
        seq = 0; nint = 0; medium = -1; skip = -1; skip_rec = -1;
        alive = true; fresh = true; walking = false; pending = false;
      }
      ring_n -= take;
    }
    ODW_GTIME(0);
    // ---- B: set up the next segment of fresh lanes ---------------------------------------------
    ODW_GSTAT(1, __ballot(alive && fresh));
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
        cut = q.tmax;
        walking = false;
        if (mask != 0ull) {
          // (one Newton step, 2^-50: the plane distances of the walk decide the order of the cells and where it ends,
          //  against boxes that carry 2 distTol of slack -- as the inverse direction of the flat kernels' box tests)
          ivx = ODW_GRID_RCP(dir.x); ivy = ODW_GRID_RCP(dir.y); ivz = ODW_GRID_RCP(dir.z);
          // A ray that goes on from a hit starts in the cell its walk stopped in: the walk ends in the cell whose
          // exit lies beyond the hit (+ 2 distTol), so the hit point is in it or within the tolerance of it.  If it
          // is a hair outside, the plane distance of that axis comes out negative (or the cell is entered at once)
          // and the first advance corrects the index: at worst one cell is looked at in vain.  A new ray brings
          // its first cell from the ring.
          const bool in = cell != -1;
          const int ix = cell & 0xff, iy = (cell >> 8) & 0xff, iz = (cell >> 16) & 0xff;
          if (in) {
            tx = dir.x > 0 ? (bx[ix + 1] - point.x) * ivx : (dir.x < 0 ? (bx[ix] - point.x) * ivx : INFINITY);
            ty = dir.y > 0 ? (bx[by_off + iy + 1] - point.y) * ivy : (dir.y < 0 ? (bx[by_off + iy] - point.y) * ivy : INFINITY);
            tz = dir.z > 0 ? (bx[bz_off + iz + 1] - point.z) * ivz : (dir.z < 0 ? (bx[bz_off + iz] - point.z) * ivz : INFINITY);
            walking = true;
          }
        }
      }
    }
    ODW_GTIME(1);
    // ---- C: cell steps, all walking lanes together -----------------------------------------------
    for (int it = 0;; ++it) {
      const uint64_t wb = __ballot(walking);
      if (wb == 0ull) break;
      // stop stepping once few lanes walk if lanes wait: for resolution / interaction, or -- enough of
      // them -- for a new ray
      // (a sorted launch whose lanes interact together: lanes that wait for the others are no reason to stop stepping,
      //  lanes that wait for their cell to be resolved are)
      if (it > 0 && __popcll(wb) < ODW_GRID_STEP_MIN &&
          (__ballot(alive && !walking && (interact_min <= 1u || pending || fresh)) != 0ull ||
           (!(drained && ring_n == 0) && (uint32_t)__popcll(__ballot(!alive)) >= refill_min)))
        break;
      ODW_GSTAT(2, wb);
      if (walking) {
        const int ix = cell & 0xff, iy = (cell >> 8) & 0xff, iz = cell >> 16;
        const int ci = ix + nx * (iy + ny * iz);
        uint32_t word;
        if (IN_LDS) word = lds32[cell_off + ci]; else word = GD.cells[ci];
        const uint32_t first = word & 0xffffffu, count = word >> 24;
        bool maybe = false;
        for (uint32_t k = 0; k < count; ++k) {
          if (SPHERES) {
            // cheap test: does the line meet the sphere (discriminant), not behind the ray
#if ODW_GRID_EXACT_WALK
            // A/B (round 4): the exact roots and consider() inside the cell step, no resolve queue -- the lanes whose
            // line meets the sphere run them under a divergent branch while the others wait
            double2 r0, r1, r2;
            if (IN_LDS) {
              const double2* rec = reinterpret_cast<const double2*>(grid_lds + item_off) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            } else {
              const double2* rec = reinterpret_cast<const double2*>(GD.items) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            }
            const uint64_t bits = (uint64_t)__double_as_longlong(r2.x);
            const int prim = (int)(uint32_t)bits, gs = (int)(uint32_t)(bits >> 32);
            const d3 oc = point - mk(r0.x, r0.y, r1.x);
            const double bh = dot(oc, dir), cc = dot(oc, oc) - r1.y * r1.y;
            if (((mask >> (gs & 0xff)) & 1) && (gs >> 8) != skip && bh * bh - cc >= 0 && (bh < 0 || cc < 0)) {
              double ta, tb;
              if (quad_roots_unit(bh, cc, ta, tb) == 2) {
                const double bt = ta > q.tol ? ta : (tb > q.tol ? tb : INFINITY);
                consider(sv, q, bt, prim, (int)(first + k), gs & 0xff, 0, 0);
                cut = fmin_raw(q.tmax, q.any.t + 2.0 * q.tol);
              }
            }
#elif ODW_GRID_LEAN_TEST
            // geometry only, from the first 32 bytes of the record (centre, radius); the sphere the ray has just left
            // is known by its record (skip_rec: the interaction keeps the record's index).  Whether the group is
            // relevant, and the solid rule for a sphere listed in a second cell, are asked when the cell is resolved
            // (phase D asks them anyway): a third less LDS traffic per step, no 64-bit shift
            double2 r0, r1;
            if (IN_LDS) {
              const double2* rec = reinterpret_cast<const double2*>(grid_lds + item_off) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1];
            } else {
              const double2* rec = reinterpret_cast<const double2*>(GD.items) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1];
            }
            const d3 oc = point - mk(r0.x, r0.y, r1.x);
            const double bh = dot(oc, dir), cc = dot(oc, oc) - r1.y * r1.y;
            maybe |= (int)(first + k) != skip_rec && bh * bh - cc >= 0 && (bh < 0 || cc < 0);
#if ODW_DOUBLE == 21
            {
              const d3 oc2 = point - mk(opq(r0.x), r0.y, r1.x);
              const double bh2 = dot(oc2, dir), cc2 = dot(oc2, oc2) - r1.y * r1.y;
              maybe |= (int)(first + k) != skip_rec && bh2 * bh2 - cc2 >= 0 && (bh2 < 0 || cc2 < 0);
            }
#endif
#else
            double2 r0, r1, r2;
            if (IN_LDS) {
              const double2* rec = reinterpret_cast<const double2*>(grid_lds + item_off) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            } else {
              const double2* rec = reinterpret_cast<const double2*>(GD.items) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            }
            const uint32_t gs = (uint32_t)((uint64_t)__double_as_longlong(r2.x) >> 32);
            const d3 oc = point - mk(r0.x, r0.y, r1.x);
            const double bh = dot(oc, dir), cc = dot(oc, oc) - r1.y * r1.y;
            // (outside the sphere and moving away from its centre: both roots negative)
            maybe |= ((mask >> (gs & 0xff)) & 1) && (int)(gs >> 8) != skip && bh * bh - cc >= 0 && (bh < 0 || cc < 0);
#endif
          } else {
            int p;
            if (IN_LDS) p = (int)lds32[2 * item_off + first + k]; else p = (int)reinterpret_cast<const uint32_t*>(GD.items)[first + k];
            // cheap test: the primitive's bounding box (tolerance slack included) against the ray up to the cut
            cf64 hdr = sv.prim_hdr + 8 * (size_t)p;
            ci32 hi = (ci32)(hdr + 6);
            const d3 oi = mk(point.x * ivx, point.y * ivy, point.z * ivz);
            maybe |= ((mask >> hi[1]) & 1) && (hi[2] >> ODW_SOLID_SHIFT) != skip &&
                     ray_box(hdr, oi, mk(ivx, ivy, ivz), fmin(q.tmax, q.any.t + 2.0 * q.tol));
          }
        }
        if (maybe) {
          walking = false;                                 // resolve this cell's primitives in phase D
          pending = true;
        } else {
          ODW_WALK_ADVANCE();
        }
      }
    }
    ODW_GTIME(2);
    // ---- D: resolution and interaction of the lanes whose walk has stopped -----------------------------
    ODW_GSTAT(3, __ballot(alive && !walking && !fresh));
    // Rays handed out in sorted order (ray_order): the lanes of a wave hold neighbouring rays; the interaction may wait
    // until interact_min lanes are through with their walk (or nobody walks or waits for a resolve any more), so that the
    // wave sets out on the next segment together (as odw_mesh_kernel does).  Index order: every lane goes on at once.
    bool go = true;
    if (SORTED && interact_min > 1u) {
      const uint64_t busy = __ballot(walking || (alive && pending));
      const uint64_t done = __ballot(alive && !walking && !fresh && !pending);
      go = busy == 0ull || (uint32_t)__popcll(done) >= interact_min;
    }
    if (alive && !walking && !fresh && (pending || go)) {
      if (pending) {
        // the exact tests of the cell the walk stands in (every primitive listed there), then on or stop
        pending = false;
        ODW_GSTAT(4, __ballot(1));
        const int ix = cell & 0xff, iy = (cell >> 8) & 0xff, iz = cell >> 16;
        const int ci = ix + nx * (iy + ny * iz);
        uint32_t word;
        if (IN_LDS) word = lds32[cell_off + ci]; else word = GD.cells[ci];
        const uint32_t first = word & 0xffffffu, count = word >> 24;
        for (uint32_t k = 0; k < count; ++k) {
          if (SPHERES) {
            double2 r0, r1, r2;
            if (IN_LDS) {
              const double2* rec = reinterpret_cast<const double2*>(grid_lds + item_off) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            } else {
              const double2* rec = reinterpret_cast<const double2*>(GD.items) + 3 * (size_t)(first + k);
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            }
            const uint64_t bits = (uint64_t)__double_as_longlong(r2.x);
            const int prim = (int)(uint32_t)bits, gs = (int)(uint32_t)(bits >> 32);
            const int g = gs & 0xff;
            if (((mask >> g) & 1) && (gs >> 8) != skip) {
              // a sphere needs no frame (as in intersect_prim): centre in global coordinates
              const d3 oc = point - mk(r0.x, r0.y, r1.x);
              double ta, tb;
#if ODW_DOUBLE == 23
              { const d3 oc2 = mk(opq(oc.x), oc.y, oc.z); double ua, ub;
                if (quad_roots_unit(dot(oc2, dir), dot(oc2, oc2) - r1.y * r1.y, ua, ub) == 2 && ua != ua) consider(sv, q, ua, prim, (int)(first + k), g, 0, 0); }
#endif
              if (quad_roots_unit(dot(oc, dir), dot(oc, oc) - r1.y * r1.y, ta, tb) == 2) {
                const double bt = ta > q.tol ? ta : (tb > q.tol ? tb : INFINITY);
                // (a sphere has one face: the candidate's face word carries the record's index instead, so that the
                //  interaction finds centre, flags and group where the cheap test found them -- hit rows hold no face)
                consider(sv, q, bt, prim, (int)(first + k), g, 0, 0);
              }
            }
          } else {
            int p;
            if (IN_LDS) p = (int)lds32[2 * item_off + first + k]; else p = (int)reinterpret_cast<const uint32_t*>(GD.items)[first + k];
            ci32 pi = sv.prim_i32 + 4 * p;
            const int g = pi[1];
            if (((mask >> g) & 1) && (pi[2] >> ODW_SOLID_SHIFT) != skip) intersect_prim(sv, q, p, pi[0], g, pi[2], pi[3]);
          }
        }
        cut = fmin_raw(q.tmax, q.any.t + 2.0 * q.tol);
        ODW_WALK_ADVANCE();
      }
      if (!walking && go) {
        ODW_GSTAT(5, __ballot(1));
        if (q.any.prim == 0x7fffffff) {
          ODW_GCOUNT(ODW_CNT_ESCAPED);
          alive = false;
        } else {
          const bool use_oth = q.oth.prim != 0x7fffffff && q.oth.t < q.any.t + 2.0 * q.tol;
          const double t_hit = use_oth ? q.oth.t : q.any.t;
          const int face = use_oth ? q.oth.face : q.any.face;
          const int prim = use_oth ? q.oth.prim : q.any.prim;
          point = point + dir * t_hit;
          if (medium >= 0) {                                  // ray.py:120-125 (assignment)
            const double L = group_f64[4 * medium + 2];
            if (L == 0) power = 0;
            else if (L < INFINITY) power = exp(-t_hit / L);
          }
          d3 n;
          int g, pflags;
          if (SPHERES) {
            // a sphere's outward normal needs no frame either: (point - centre) / |.|; centre, group and flag word
            // from the record of the cheap test (`face` = its index): no read of the primitive tables
            double2 r0, r1, r2;
            if (IN_LDS) {
              const double2* rec = reinterpret_cast<const double2*>(grid_lds + item_off) + 3 * (size_t)face;
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            } else {
              const double2* rec = reinterpret_cast<const double2*>(GD.items) + 3 * (size_t)face;
              r0 = rec[0]; r1 = rec[1]; r2 = rec[2];
            }
            const d3 v = point - mk(r0.x, r0.y, r1.x);
            n = v * frsqrt(dot(v, v));
            g = (int)(((uint64_t)__double_as_longlong(r2.x) >> 32) & 0xff);
            pflags = (int)(uint32_t)(uint64_t)__double_as_longlong(r2.y);
            if (pflags & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
          } else {
            cf64 pf = sv.prim_f64 + (size_t)prim * 16;
            ci32 pi = sv.prim_i32 + 4 * prim;
            n = face_normal(pi[0], pf + 12, face, xf_point(pf, point));
            g = pi[1];
            pflags = pi[2];
            if (pflags & ODW_FLAG_FLIP_NORMAL) n = n * -1.0;
            n = xf_vec_t(pf, n);
          }
          const bool entering = dot(dir, n) < 0;
          if (entering) n = n * -1.0;
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
          skip = ((pflags & ODW_FLAG_CONVEX) && (entering ? -dot(dir, n) : dot(dir, n)) > 0) ? (pflags >> ODW_SOLID_SHIFT) : -1;
          skip_rec = (SPHERES && skip >= 0) ? face : -1;
          if (alive && power < lim.power_tol) { ODW_GCOUNT(ODW_CNT_DIED); alive = false; }
          fresh = alive;
        }
        if (!alive) {
          atomicAdd(&wave_cnt[ODW_CNT_SEGMENTS], (uint32_t)nint);
          ODW_GCOUNT(ODW_CNT_TRACED_RAYS);
        }
      }
    }
    ODW_GTIME(3);
  }
#undef ODW_WALK_ADVANCE
#undef refill_min
#undef interact_min
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
#ifdef ODW_GRID_STATS
  if (lane < 12 && P.dbg) atomicAdd(P.dbg + lane, (unsigned long long)wave_cnt[8 + lane]);
  if (lane == 0 && P.dbg)
    for (int k = 0; k < 4; ++k) atomicAdd(P.dbg + 24 + k, (unsigned long long)phase_t[k]);
#endif
  // (one atomic per block and counter: the 16 waves' words are next to each other in LDS)
  __syncthreads();
  if (threadIdx.x < ODW_CNT_LDS) {
    uint32_t s = 0;
    for (int w = 0; w < ODW_GRID_WAVES; ++w) s += lds32[word_off + w * ODW_GRID_WAVE_WORDS + threadIdx.x];
    if (s) atomicAdd(P.out.counters + threadIdx.x, (unsigned long long)s);
  }
}

#if ODW_GRID_SORTED
// ---- the order rays are handed out in (TraceParams.ray_order), grid launches ------------------------------------------
// key of ray r = the two uniform numbers its direction is drawn from (azimuth and polar inverse CDFs are monotone in
// them), 16 bits each, Morton order: Philox alone, no table inversion, no sin / cos.  Sorted by its top bits, the rays of
// a wave leave the source side by side (equal-probability patches of the beam).
__device__ __forceinline__ uint32_t grid_spread16(uint32_t v) {        // abcd -> 0a0b0c0d
  v &= 0xffffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__global__ __launch_bounds__(256) void odw_ray_ukey_kernel(uint64_t first, uint64_t n, uint64_t seed,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t ray = first + r;
  uint32_t c0 = (uint32_t)ray, c1 = (uint32_t)(ray >> 32), c2 = 0u, c3 = 0u;
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));      // (ray_uniforms: u_phi from c0, u_t from c2)
  keys[r] = grid_spread16(c0 >> 16) | (grid_spread16(c2 >> 16) << 1);
  vals[r] = (uint32_t)r;
}
#endif

}  // namespace odw
