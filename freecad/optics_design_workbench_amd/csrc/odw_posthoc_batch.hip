// odw_posthoc_batch.hip -- the post-hoc steps of odw_posthoc.hip for ALL segments of a batch launch (odw_trace_batch),
// with every decision between two kernels taken ON THE DEVICE.
//
// One scene's chain -- select, sample, [plane search on the host], project + medians, [origin], bin -- cost eight
// synchronisations with the host in round 3; round 4 ran every step for all S segments and waited once per step (eight
// waits per group).  Round 5: the per-scene state (row counts, extrema, the bins that hold the middle ranks, the median
// candidates, the origin) lives in a device-resident record (PhbScene); tiny kernels take the decisions the host took
// (which histogram bins hold the ranks: ph_rank_bins; the ranks among the collected candidates: std::sort + index); kernels
// are launched ONCE for all scenes (blockIdx.y = scene).  A group's chain is two stream-ordered pieces with ONE hand-over to
// the host between them (the plane search on the <= 300-row sample, numpy's arithmetic):
//   begin    popcounts of the row-of-ray table -> prefix sums -> per-scene state -> rows by rank -> the thinned sample
//            -> sample + state to page-locked host memory, event
//   measure  planes from the host -> projection + extrema + moments -> coarse histogram -> decide -> fine histogram ->
//            decide -> collect -> pick the middle elements, origin -> bin -> state + moment sums + counts to the host, event
// odw_batch_hits_begin / _sampled / _measure / _measured enqueue and poll; the older one-wait-per-step entry points
// (odw_batch_hits_select / sample / project / bin) are the same kernels with a wait after each piece.  Scene by scene the
// results are those of the per-segment calls (odw_hits_*), bit for bit: medians by exact selection, moments with the same
// grid and order of additions, counts are integers.
// Included by odw_capi.hip (one translation unit, after odw_posthoc.hip).

namespace {

constexpr uint64_t kPhbRowsRoom = 4104, kPhbBinsRoom = 4096;   // sample rows / histogram bins per scene the buffers hold from the start
constexpr uint32_t kPhbCand = 2048;        // median candidates per coordinate the pick kernel ranks; more (a cloud piled up on
                                           // one value): that coordinate is sorted (sync entry points) / the scene reported
enum : uint32_t { PHB_SLOW_X = 1u, PHB_SLOW_Y = 2u, PHB_BAD_HIST = 4u, PHB_SAMPLE_CUT = 8u };

struct PhbScene {
  uint64_t m;              // rows selected (one per ray that has a row)
  uint64_t leaving;        // ... of rays that leave
  uint64_t used;           // slots of the segment in use
  uint64_t sample_stride;
  uint32_t sample_n;
  uint32_t ordered;        // the ordered selection without a sort serves this scene
  uint32_t on;             // projection / binning include it (ordered, m > 0, not skipped by the caller)
  uint32_t flags;          // PHB_*
  double ex[3], ey[3];     // in-plane axes (from the host's plane search)
  double ext[4];           // min X, max X, min Y, max Y
  PhSel sel;
  uint64_t k_lo, k_hi;     // the middle ranks (numpy.median)
  uint64_t cbelow[2][2], fbelow[2][2];
  uint32_t cbin[2][2], fbin[2][2];
  uint32_t take[2], n0[2]; // candidates per coordinate, and those of the first fine bin
  unsigned long long n_cand[2];
  double stats[8];         // as odw_hits_project
  double origin[2];        // numpy.median of X, Y
};

__global__ __launch_bounds__(256) void phb_mark_kernel(const odw_hit* __restrict__ hits_all, uint64_t slots, const unsigned long long* __restrict__ hit_count,
                                                       int group, uint64_t ray0, uint64_t n_rays, uint64_t rays_pad, uint32_t* __restrict__ row_of_all,
                                                       unsigned long long* __restrict__ small_all) {
  const int s = blockIdx.y;
  const odw_hit* hits = hits_all + (size_t)s * slots;
  uint32_t* row_of = row_of_all + (size_t)s * rays_pad;
  unsigned long long* counts = small_all + 4 * (size_t)s;
  const uint64_t n = hit_count[4 * s] < slots ? hit_count[4 * s] : slots;
  uint32_t n_sel = 0, n_leave = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t tag = hits[i].tag;
    const bool sel = tag != ODW_TAG_UNUSED && (group < 0 || (int)ODW_HIT_GROUP(tag) == group);
    if (sel) {
      const uint64_t r = ODW_HIT_RAY(tag) - ray0;
      if (r >= n_rays) { *(uint32_t*)(counts + 3) = 1u; continue; }
      row_of[r] = (uint32_t)i;
      n_sel += 1u;
      n_leave += ODW_HIT_ENTERING(tag) ? 0u : 1u;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n_sel += __shfl_xor(n_sel, off); n_leave += __shfl_xor(n_leave, off); }
  __shared__ uint32_t sh[4][2];
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = n_sel; sh[threadIdx.x >> 6][1] = n_leave; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0], b = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
    if (a) atomicAdd(counts, (unsigned long long)a);
    if (b) atomicAdd(counts + 1, (unsigned long long)b);
  }
}

// (ph_popc_kernel for every scene: words = [bitmap | popcounts | prefix sums] per scene)
// kPhbPer rays per thread, their loads issued together: with one 4-byte load per thread the pass was bound by the latency of
// that load (40 MB in 22 us), not by bandwidth
constexpr int kPhbPer = 8;
__global__ __launch_bounds__(256) void phb_popc_kernel(const uint32_t* __restrict__ row_of_all, uint64_t n_rays, uint64_t rays_pad,
                                                       uint32_t* __restrict__ words_all, uint64_t n_words) {
  const uint32_t* row_of = row_of_all + (size_t)blockIdx.y * rays_pad;
  uint32_t* bitmap = words_all + (size_t)blockIdx.y * 3 * n_words;
  uint32_t* pop = bitmap + n_words;
  const uint64_t base = (uint64_t)blockIdx.x * (blockDim.x * kPhbPer) + threadIdx.x;
  uint32_t v[kPhbPer];
#pragma unroll
  for (int k = 0; k < kPhbPer; ++k) {
    const uint64_t r = base + (uint64_t)k * blockDim.x;
    v[k] = r < n_rays ? row_of[r] : PH_NO_ROW;
  }
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int k = 0; k < kPhbPer; ++k) {
    const uint64_t r = base + (uint64_t)k * blockDim.x;
    const uint64_t b = __ballot(v[k] != PH_NO_ROW);
    if ((lane & 31u) == 0 && r < n_rays) {
      const uint32_t word = (uint32_t)(lane ? b >> 32 : b);
      bitmap[r >> 5] = word;
      pop[r >> 5] = (uint32_t)__popc(word);
    }
  }
}

// what the host worked out after odw_batch_hits_select's first wait: rows, leaving rows, whether the ordered selection
// without a sort serves the scene; the stride and size of the thinned sample
__global__ void phb_state_kernel(PhbScene* __restrict__ scenes, int S, const unsigned long long* __restrict__ hit_count,
                                 const unsigned long long* __restrict__ small_all, const uint32_t* __restrict__ words_all, uint64_t n_words,
                                 uint64_t n_rays, uint64_t slots, int marked, uint64_t limit, uint64_t cap) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const unsigned long long* c = hit_count + 4 * (size_t)s;
  const uint64_t used = c[0] < slots ? c[0] : slots;
  const uint64_t rows_s = used > c[1] ? used - c[1] : 0;
  const bool dropped = c[0] > slots;
  const uint32_t* bitmap = words_all + (size_t)s * 3 * n_words;
  const uint32_t* before = bitmap + 2 * n_words;
  const uint64_t w = (n_rays - 1) >> 5;
  const uint64_t marked_rays = (uint64_t)before[w] + (uint64_t)__popc(bitmap[w]);
  const unsigned long long* sm = small_all + 4 * (size_t)s;
  // (marked by the launch: rows and leaving rows by its own counters, dropped rows never reached the table)
  const uint64_t sel = marked ? rows_s : sm[0], leave = marked ? (dropped ? 0 : c[2]) : sm[1];
  const uint64_t oob = marked ? 0 : sm[3];
  // (two rows of one ray: one store of the two stands, fewer rays marked than rows selected; a sample of the entering
  //  rows of a list that also holds leaving ones needs a compaction of its own: both take the per-segment calls)
  const bool mixed = leave > 0 && (double)leave < .51 * (double)sel;
  PhbScene& P = scenes[s];
  P.m = sel;
  P.leaving = leave;
  P.used = used;
  P.ordered = (oob == 0 && marked_rays == sel && !mixed && !(marked && dropped)) ? 1u : 0u;
  P.on = 0;
  P.flags = 0;
  const uint64_t stride = 1 + (limit ? sel / limit : 0);
  uint64_t count = (P.ordered && sel) ? (sel + stride - 1) / stride : 0;
  if (count > cap) { count = cap; P.flags |= PHB_SAMPLE_CUT; }
  P.sample_stride = stride;
  P.sample_n = (uint32_t)count;
}

__global__ __launch_bounds__(256) void phb_rank_kernel(const uint32_t* __restrict__ words_all, uint64_t n_words, const uint32_t* __restrict__ row_of_all,
                                                       uint64_t n_rays, uint64_t rays_pad, uint32_t* __restrict__ sel_all, uint64_t slots) {
  const uint32_t* bitmap = words_all + (size_t)blockIdx.y * 3 * n_words;
  const uint32_t* before = bitmap + 2 * n_words;
  const uint32_t* row_of = row_of_all + (size_t)blockIdx.y * rays_pad;
  uint32_t* out = sel_all + (size_t)blockIdx.y * slots;
  const uint64_t base = (uint64_t)blockIdx.x * (blockDim.x * kPhbPer) + threadIdx.x;
  uint32_t word[kPhbPer], bef[kPhbPer], row[kPhbPer];
#pragma unroll
  for (int k = 0; k < kPhbPer; ++k) {
    const uint64_t r = base + (uint64_t)k * blockDim.x;
    const bool in = r < n_rays;
    word[k] = in ? bitmap[r >> 5] : 0u;
    bef[k] = in ? before[r >> 5] : 0u;
    row[k] = in ? row_of[r] : PH_NO_ROW;
  }
#pragma unroll
  for (int k = 0; k < kPhbPer; ++k) {
    const uint64_t r = base + (uint64_t)k * blockDim.x;
    const uint32_t bit = 1u << (r & 31u);
    if (r < n_rays && (word[k] & bit)) {
      const uint64_t at = (uint64_t)bef[k] + (uint32_t)__popc(word[k] & (bit - 1u));
      if (at < slots) out[at] = row[k];
    }
  }
}

// rows sel[j * stride], j < count, of every ordered scene: four lanes per 64-byte row.  strides: per scene (the caller's),
// or null: the scene's own (points[::1 + n / limit])
__global__ void phb_sample_kernel(const PhbScene* __restrict__ scenes, const uint64_t* __restrict__ strides, const odw_hit* __restrict__ hits_all,
                                  uint64_t slots, const uint32_t* __restrict__ sel_all, odw_hit* __restrict__ out_all, uint64_t cap,
                                  uint64_t keep) {
  const PhbScene& P = scenes[blockIdx.y];
  if (!P.ordered || P.m == 0) return;
  uint64_t stride = P.sample_stride, count = P.sample_n;
  if (keep) {                     // rows [::max(1, m // keep)]
    stride = P.m / keep ? P.m / keep : 1;
    count = (P.m + stride - 1) / stride;
    if (count > cap) count = cap;
  } else if (strides) {
    stride = strides[blockIdx.y] ? strides[blockIdx.y] : 1;
    count = (P.m + stride - 1) / stride;
    if (count > cap) count = cap;
  }
  const odw_hit* hits = hits_all + (size_t)blockIdx.y * slots;
  const uint32_t* sel = sel_all + (size_t)blockIdx.y * slots;
  odw_hit* out = out_all + (size_t)blockIdx.y * cap;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = t >> 2;
  if (row < count) {
    const double2* src = reinterpret_cast<const double2*>(hits + sel[row * stride]);
    reinterpret_cast<double2*>(out + row)[t & 3] = src[t & 3];
  }
}

// the caller's planes: [S][7] = ex (3), ey (3), skip
__global__ void phb_planes_kernel(PhbScene* __restrict__ scenes, int S, const double* __restrict__ planes) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  PhbScene& P = scenes[s];
  for (int k = 0; k < 3; ++k) { P.ex[k] = planes[7 * s + k]; P.ey[k] = planes[7 * s + 3 + k]; }
  P.on = (P.ordered && P.m > 0 && planes[7 * s + 6] == 0.0) ? 1u : 0u;
  P.flags &= PHB_SAMPLE_CUT;
}
__global__ void phb_origins_kernel(PhbScene* __restrict__ scenes, int S, const double* __restrict__ origins) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  // (the caller's origin stands: coordinates the pick kernel could not serve were sorted by the caller's entry point)
  if (s < S) { scenes[s].origin[0] = origins[2 * s]; scenes[s].origin[1] = origins[2 * s + 1]; scenes[s].flags &= ~(PHB_SLOW_X | PHB_SLOW_Y); }
}

// ph_project_kernel (key = points) for every scene that is on; a scene of m rows uses the blocks the per-segment launch
// would (min(ceil(m / 256), gridDim.x)): the same strides, the same order of additions in the moment sums
__global__ __launch_bounds__(256) void phb_project_kernel(const PhbScene* __restrict__ scenes, const odw_hit* __restrict__ hits_all, uint64_t slots,
                                                          const double* __restrict__ pts_all, const uint32_t* __restrict__ sel_all, double* __restrict__ X_all,
                                                          double* __restrict__ Y_all, uint64_t xy_stride, double* __restrict__ part_all,
                                                          uint64_t part_stride) {
#pragma clang fp contract(off)
  const PhbScene& P = scenes[blockIdx.y];
  if (!P.on) return;
  const uint64_t m = P.m;
  const uint64_t want = (m + 255) / 256;
  const unsigned g = (unsigned)(want < (uint64_t)gridDim.x ? (want ? want : 1) : gridDim.x);
  if (blockIdx.x >= g) return;
  const odw_hit* hits = hits_all + (size_t)blockIdx.y * slots;
  const uint32_t* sel = sel_all + (size_t)blockIdx.y * slots;
  double* X = X_all + (size_t)blockIdx.y * xy_stride;
  double* Y = Y_all + (size_t)blockIdx.y * xy_stride;
  double* part = part_all + (size_t)blockIdx.y * part_stride;
  const double ex0 = P.ex[0], ex1 = P.ex[1], ex2 = P.ex[2], ey0 = P.ey[0], ey1 = P.ey[1], ey2 = P.ey[2];
  double lo_x = INFINITY, hi_x = -INFINITY, lo_y = INFINITY, hi_y = -INFINITY;
  // (the points by themselves where the launch wrote them too: 24 bytes per row instead of the row's 64)
  const double* pts = pts_all ? pts_all + (size_t)blockIdx.y * slots * 3 : nullptr;
  const double cx = pts ? pts[sel[0]] : hits[sel[0]].point[0], cy = pts ? pts[slots + sel[0]] : hits[sel[0]].point[1],
               cz = pts ? pts[2 * slots + sel[0]] : hits[sel[0]].point[2];
  double mom[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t stride = (uint64_t)g * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    const uint32_t at = sel[j];
    const double a = pts ? pts[at] : hits[at].point[0], b = pts ? pts[slots + at] : hits[at].point[1],
                 c = pts ? pts[2 * slots + at] : hits[at].point[2];
    const double x = a * ex0 + b * ex1 + c * ex2, y = a * ey0 + b * ey1 + c * ey2;
    X[j] = x;
    Y[j] = y;
    lo_x = fmin(lo_x, x); hi_x = fmax(hi_x, x);
    lo_y = fmin(lo_y, y); hi_y = fmax(hi_y, y);
    ph_moment_add(mom, a, b, c, cx, cy, cz);
  }
  ph_moment_write(mom, cx, cy, cz, part + 4 * (size_t)g, g);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = fmin(lo_x, __shfl_xor(lo_x, off)); hi_x = fmax(hi_x, __shfl_xor(hi_x, off));
    lo_y = fmin(lo_y, __shfl_xor(lo_y, off)); hi_y = fmax(hi_y, __shfl_xor(hi_y, off));
  }
  __shared__ double s[4][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[w][0] = lo_x; s[w][1] = hi_x; s[w][2] = lo_y; s[w][3] = hi_y; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      s[0][0] = fmin(s[0][0], s[k][0]); s[0][1] = fmax(s[0][1], s[k][1]);
      s[0][2] = fmin(s[0][2], s[k][2]); s[0][3] = fmax(s[0][3], s[k][3]);
    }
    for (int k = 0; k < 4; ++k) part[4 * blockIdx.x + k] = s[0][k];
  }
}

// extrema of a scene's projection from its blocks' (one block per scene), and what ph_select_stats sets up from them
__global__ __launch_bounds__(256) void phb_ext_kernel(PhbScene* __restrict__ scenes, const double* __restrict__ part_all, uint64_t part_stride,
                                                      unsigned gmax) {
  PhbScene& P = scenes[blockIdx.x];
  if (!P.on) return;
  const uint64_t want = (P.m + 255) / 256;
  const unsigned g = (unsigned)(want < (uint64_t)gmax ? (want ? want : 1) : gmax);
  const double* part = part_all + (size_t)blockIdx.x * part_stride;
  double e0 = INFINITY, e1 = -INFINITY, e2 = INFINITY, e3 = -INFINITY;
  for (unsigned b = threadIdx.x; b < g; b += blockDim.x) {
    e0 = fmin(e0, part[4 * b]); e1 = fmax(e1, part[4 * b + 1]);
    e2 = fmin(e2, part[4 * b + 2]); e3 = fmax(e3, part[4 * b + 3]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    e0 = fmin(e0, __shfl_xor(e0, off)); e1 = fmax(e1, __shfl_xor(e1, off));
    e2 = fmin(e2, __shfl_xor(e2, off)); e3 = fmax(e3, __shfl_xor(e3, off));
  }
  __shared__ double s[4][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[w][0] = e0; s[w][1] = e1; s[w][2] = e2; s[w][3] = e3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      s[0][0] = fmin(s[0][0], s[k][0]); s[0][1] = fmax(s[0][1], s[k][1]);
      s[0][2] = fmin(s[0][2], s[k][2]); s[0][3] = fmax(s[0][3], s[k][3]);
    }
    PhSel sel;
    for (int a = 0; a < 2; ++a) {
      P.ext[2 * a] = s[0][2 * a]; P.ext[2 * a + 1] = s[0][2 * a + 1];
      const double width = s[0][2 * a + 1] - s[0][2 * a];
      sel.lo[a] = s[0][2 * a];
      sel.sc[a] = (width > 0 && width < INFINITY) ? (double)kPhSelBins / width : 0.0;
      sel.c_lo[a] = 0; sel.c_hi[a] = 0; sel.flo[a] = 0; sel.fsc[a] = 0; sel.f0[a] = 0; sel.f1[a] = 0;
      P.n_cand[a] = 0; P.take[a] = 0; P.n0[a] = 0;
    }
    sel.fine = 0;
    P.sel = sel;
    P.k_lo = (P.m - 1) / 2;
    P.k_hi = P.m / 2;
    P.stats[2] = s[0][0]; P.stats[3] = s[0][1]; P.stats[6] = s[0][2]; P.stats[7] = s[0][3];
  }
}

// ph_sel_hist_kernel for every scene that is on (slices: [S][kPhSelBlocks][2 kPhSelBins])
__global__ __launch_bounds__(512) void phb_sel_hist_kernel(const PhbScene* __restrict__ scenes, const double* __restrict__ X_all,
                                                           const double* __restrict__ Y_all, uint64_t xy_stride, uint32_t* __restrict__ slices_all) {
  const PhbScene& S = scenes[blockIdx.y];
  if (!S.on) return;
  const uint64_t m = S.m;
  const uint64_t want = (m + 511) / 512;
  const unsigned g = (unsigned)(want < (uint64_t)gridDim.x ? (want ? want : 1) : gridDim.x);
  if (blockIdx.x >= g) return;
  const PhSel P = S.sel;
  const double* X = X_all + (size_t)blockIdx.y * xy_stride;
  const double* Y = Y_all + (size_t)blockIdx.y * xy_stride;
  __shared__ uint32_t h[2 * kPhSelBins];
  for (int k = threadIdx.x; k < 2 * kPhSelBins; k += blockDim.x) h[k] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)g * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    const double v[2] = {X[j], Y[j]};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const uint32_t c = ph_sel_bin(v[a], P.lo[a], P.sc[a]);
      if (!P.fine) atomicAdd(&h[a * kPhSelBins + c], 1u);
      else if (c >= P.c_lo[a] && c <= P.c_hi[a]) atomicAdd(&h[a * kPhSelBins + ph_sel_bin(v[a], P.flo[a], P.fsc[a])], 1u);
    }
  }
  __syncthreads();
  uint32_t* out = slices_all + ((size_t)blockIdx.y * kPhSelBlocks + blockIdx.x) * 2 * kPhSelBins;
  for (int k = threadIdx.x; k < 2 * kPhSelBins; k += blockDim.x) out[k] = h[k];
}
__global__ __launch_bounds__(256) void phb_sel_sum_kernel(const PhbScene* __restrict__ scenes, const uint32_t* __restrict__ slices_all,
                                                          uint32_t* __restrict__ hist_all) {
  const PhbScene& S = scenes[blockIdx.z];
  if (!S.on) return;
  const uint64_t want = (S.m + 511) / 512;
  const int n_slices = (int)(want < (uint64_t)kPhSelBlocks ? (want ? want : 1) : kPhSelBlocks);
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < 2 * kPhSelBins) {
    const int b0 = blockIdx.y * kPhSelSumSlices, b1 = min(n_slices, b0 + kPhSelSumSlices);
    const uint32_t* slices = slices_all + (size_t)blockIdx.z * kPhSelBlocks * 2 * kPhSelBins;
    uint32_t s = 0;
    for (int b = b0; b < b1; ++b) s += slices[(size_t)b * 2 * kPhSelBins + k];
    if (s) atomicAdd(hist_all + (size_t)blockIdx.z * 2 * kPhSelBins + k, s);
  }
}

// ph_rank_bins by one block of 256 threads: the bins that hold ranks k0 <= k1 in a histogram of kPhSelBins bins, and the
// elements below each (the first bin whose running sum exceeds the rank)
__device__ bool phb_rank_bins(const uint32_t* __restrict__ h, uint64_t k0, uint64_t k1, uint32_t bin[2], uint64_t below[2],
                              uint64_t* scan /* [256] shared */, uint64_t* res /* [4] shared */, uint32_t* found /* [2] shared */) {
  constexpr int per = kPhSelBins / 256;
  const int t = threadIdx.x;
  uint64_t own = 0;
  for (int b = 0; b < per; ++b) own += h[t * per + b];
  scan[t] = own;
  if (t < 2) found[t] = 0;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {          // inclusive prefix sums
    const uint64_t add = t >= off ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += add;
    __syncthreads();
  }
  const uint64_t excl = scan[t] - own;
  for (int q = 0; q < 2; ++q) {
    const uint64_t k = q ? k1 : k0;
    if (k >= excl && k < excl + own) {
      uint64_t run = excl;
      for (int b = 0; b < per; ++b) {
        const uint64_t next = run + h[t * per + b];
        if (k < next) { res[2 * q] = (uint64_t)(t * per + b); res[2 * q + 1] = run; found[q] = 1; break; }
        run = next;
      }
    }
  }
  __syncthreads();
  const bool ok = found[0] && found[1];
  bin[0] = (uint32_t)res[0]; below[0] = res[1];
  bin[1] = (uint32_t)res[2]; below[1] = res[3];
  __syncthreads();
  return ok;
}

// after the coarse histogram: the coarse bins of the middle ranks, the fine bins over their value range
__global__ __launch_bounds__(256) void phb_decide1_kernel(PhbScene* __restrict__ scenes, const uint32_t* __restrict__ hist_all) {
#pragma clang fp contract(off)
  PhbScene& P = scenes[blockIdx.x];
  if (!P.on) return;
  __shared__ uint64_t scan[256], res[4];
  __shared__ uint32_t found[2];
  const uint32_t* hist = hist_all + (size_t)blockIdx.x * 2 * kPhSelBins;
  for (int a = 0; a < 2; ++a) {
    uint32_t bin[2];
    uint64_t below[2];
    const bool ok = phb_rank_bins(hist + (size_t)a * kPhSelBins, P.k_lo, P.k_hi, bin, below, scan, res, found);
    if (threadIdx.x == 0) {
      if (!ok) { P.flags |= PHB_BAD_HIST; bin[0] = bin[1] = 0; below[0] = below[1] = 0; }
      P.cbin[a][0] = bin[0]; P.cbin[a][1] = bin[1];
      P.cbelow[a][0] = below[0]; P.cbelow[a][1] = below[1];
      P.sel.c_lo[a] = bin[0]; P.sel.c_hi[a] = bin[1];
      const double width = P.ext[2 * a + 1] - P.ext[2 * a];
      const double w = width / (double)kPhSelBins;
      P.sel.flo[a] = P.sel.lo[a] + (double)bin[0] * w;
      const double fw = (double)(bin[1] - bin[0] + 1) * w;
      P.sel.fsc[a] = (fw > 0 && fw < INFINITY) ? (double)kPhSelBins / fw : 0.0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) P.sel.fine = 1;
}
// after the fine histogram: the fine bins of the middle ranks, how many elements they hold
__global__ __launch_bounds__(256) void phb_decide2_kernel(PhbScene* __restrict__ scenes, const uint32_t* __restrict__ hist_all) {
  PhbScene& P = scenes[blockIdx.x];
  if (!P.on) return;
  __shared__ uint64_t scan[256], res[4];
  __shared__ uint32_t found[2];
  const uint32_t* hist = hist_all + (size_t)blockIdx.x * 2 * kPhSelBins;
  for (int a = 0; a < 2; ++a) {
    uint32_t bin[2];
    uint64_t below[2];
    const uint32_t* h = hist + (size_t)a * kPhSelBins;
    const bool ok = phb_rank_bins(h, P.k_lo - P.cbelow[a][0], P.k_hi - P.cbelow[a][0], bin, below, scan, res, found);
    if (threadIdx.x == 0) {
      if (!ok) { P.flags |= PHB_BAD_HIST; bin[0] = bin[1] = 0; below[0] = below[1] = 0; }
      P.fbin[a][0] = bin[0]; P.fbin[a][1] = bin[1];
      P.fbelow[a][0] = below[0]; P.fbelow[a][1] = below[1];
      P.sel.f0[a] = bin[0]; P.sel.f1[a] = bin[1];
      const uint64_t take = (uint64_t)h[bin[0]] + (bin[1] != bin[0] ? (uint64_t)h[bin[1]] : 0ull);
      P.n0[a] = h[bin[0]];
      if (take > kPhbCand || !ok) {
        P.flags |= a ? PHB_SLOW_Y : PHB_SLOW_X;
        P.take[a] = 0;
        P.sel.c_lo[a] = 1; P.sel.c_hi[a] = 0;          // (collects nothing)
      } else {
        P.take[a] = (uint32_t)take;
      }
    }
    __syncthreads();
  }
}

// ph_sel_collect_kernel for every scene that is on: candidates of X at cand[(2 s) kPhbCand ...], of Y at [(2 s + 1) kPhbCand ...]
__global__ __launch_bounds__(256) void phb_collect_kernel(PhbScene* __restrict__ scenes, const double* __restrict__ X_all, const double* __restrict__ Y_all,
                                                          uint64_t xy_stride, double* __restrict__ cand_all) {
  PhbScene& S = scenes[blockIdx.y];
  if (!S.on) return;
  const uint64_t m = S.m;
  const PhSel P = S.sel;
  const double* X = X_all + (size_t)blockIdx.y * xy_stride;
  const double* Y = Y_all + (size_t)blockIdx.y * xy_stride;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t rounds = (m + stride - 1) / stride;
  const int lane = __lane_id();
  for (uint64_t r = 0; r < rounds; ++r) {
    const uint64_t j = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double v[2] = {j < m ? X[j] : 0.0, j < m ? Y[j] : 0.0};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      bool want = false;
      if (j < m) {
        const uint32_t c = ph_sel_bin(v[a], P.lo[a], P.sc[a]);
        if (c >= P.c_lo[a] && c <= P.c_hi[a]) {
          const uint32_t f = ph_sel_bin(v[a], P.flo[a], P.fsc[a]);
          want = f == P.f0[a] || f == P.f1[a];
        }
      }
      const unsigned long long b = __ballot(want);
      if (b) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&S.n_cand[a], (unsigned long long)__popcll(b));
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
        const unsigned long long at = base + __popcll(b & ((1ull << lane) - 1ull));
        if (want && at < kPhbCand) cand_all[((size_t)blockIdx.y * 2 + a) * kPhbCand + at] = v[a];
      }
    }
  }
}

// the two middle elements among a coordinate's candidates (what std::sort + index gave the host): candidate v stands at
// sorted positions [#smaller, #smaller + #equal); then numpy.median's mean of the two.  grid (2, S)
__global__ __launch_bounds__(256) void phb_pick_kernel(PhbScene* __restrict__ scenes, const double* __restrict__ cand_all) {
  PhbScene& P = scenes[blockIdx.y];
  if (!P.on) return;
  const int a = blockIdx.x;
  if (P.flags & (a ? PHB_SLOW_Y : PHB_SLOW_X)) return;
  __shared__ double c[kPhbCand];
  __shared__ double mid[2];
  __shared__ uint32_t got[2];
  const uint32_t take = P.take[a];
  const double* src = cand_all + ((size_t)blockIdx.y * 2 + a) * kPhbCand;
  for (uint32_t i = threadIdx.x; i < take; i += blockDim.x) c[i] = src[i];
  if (threadIdx.x < 2) got[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t r_lo = P.k_lo - P.cbelow[a][0], r_hi = P.k_hi - P.cbelow[a][0];
  const uint64_t i_lo = r_lo - P.fbelow[a][0];
  const uint64_t i_hi = P.fbin[a][1] == P.fbin[a][0] ? r_hi - P.fbelow[a][0] : (uint64_t)P.n0[a] + (r_hi - P.fbelow[a][1]);
  for (uint32_t i = threadIdx.x; i < take; i += blockDim.x) {
    const double v = c[i];
    uint32_t less = 0, eq = 0;
    for (uint32_t j = 0; j < take; ++j) { less += c[j] < v ? 1u : 0u; eq += c[j] == v ? 1u : 0u; }
    if (i_lo >= less && i_lo < (uint64_t)less + eq) { mid[0] = v; got[0] = 1; }
    if (i_hi >= less && i_hi < (uint64_t)less + eq) { mid[1] = v; got[1] = 1; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!got[0] || !got[1] || P.n_cand[a] != (unsigned long long)take) {
      P.flags |= PHB_BAD_HIST;
    } else {
      P.stats[4 * a] = mid[0];
      P.stats[4 * a + 1] = mid[1];
      P.origin[a] = (mid[0] + mid[1]) / 2.0;        // numpy.mean of the two middle elements
    }
  }
}

template <bool LDS_COUNTS>
__global__ __launch_bounds__(256) void phb_bin_kernel(const PhbScene* __restrict__ scenes, const double* __restrict__ X_all,
                                                      const double* __restrict__ Y_all, uint64_t xy_stride, int polar,
                                                      const double* __restrict__ edges_a, int na, const double* __restrict__ edges_b, int nb,
                                                      const PhbBinAccel* __restrict__ accel, unsigned long long* __restrict__ counts_all) {
#pragma clang fp contract(off)
  const PhbScene& P = scenes[blockIdx.y];
  if (!P.on || (P.flags & (PHB_SLOW_X | PHB_SLOW_Y | PHB_BAD_HIST))) return;
  const uint64_t m = P.m;
  const double ox = P.origin[0], oy = P.origin[1];
  const double* X = X_all + (size_t)blockIdx.y * xy_stride;
  const double* Y = Y_all + (size_t)blockIdx.y * xy_stride;
  const int nbins = (na - 1) * (nb - 1);
  unsigned long long* counts = counts_all + (size_t)blockIdx.y * nbins;
  // dynamic LDS, sized by the caller to what this histogram needs ([edges | counts | guide]: ph_bin_lds_bytes) -- fixed arrays
  // for the largest case (56 KB) left room for two blocks per CU, and the kernel waits on dependent LDS reads
  extern __shared__ double ph_bin_lds[];
  const bool edges_in_lds = na + nb <= kPhLdsEdges;
  double* s_edges = ph_bin_lds;
  uint32_t* s_counts = reinterpret_cast<uint32_t*>(ph_bin_lds + (edges_in_lds ? na + nb : 0));
  const bool az_fast = polar && accel->az_on && edges_in_lds;
  const bool r_fast = polar && accel->r_on && edges_in_lds;
  uint16_t* s_guide = reinterpret_cast<uint16_t*>(s_counts + (LDS_COUNTS ? nbins : 0));
  if (edges_in_lds) {
    for (int k = threadIdx.x; k < na; k += blockDim.x) s_edges[k] = edges_a[k];
    for (int k = threadIdx.x; k < nb; k += blockDim.x) s_edges[na + k] = edges_b[k];
  }
  if (LDS_COUNTS)
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) s_counts[k] = 0;
  if (r_fast)
    for (int k = threadIdx.x; k < kPhbGuide + 2; k += blockDim.x) s_guide[k] = accel->guide[k];
  __syncthreads();
  const double* ea = edges_in_lds ? s_edges : edges_a;
  const double* eb = edges_in_lds ? s_edges + na : edges_b;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // (two or four rows per trip, `#pragma unroll`: 556 -> 648 us per 8 x 1e7 rows -- the kernel is bound by its arithmetic)
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    int ia, ib;
    ph_bin_pair(accel, az_fast, r_fast, s_guide, polar, X[j] - ox, Y[j] - oy, ea, na, eb, nb, ia, ib);
    if (ia >= 0 && ib >= 0) {
      const int k = ia * (nb - 1) + ib;
      if (LDS_COUNTS) atomicAdd(&s_counts[k], 1u);
      else atomicAdd(counts + k, 1ull);
    }
  }
  if (LDS_COUNTS) {
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += blockDim.x)
      if (s_counts[k]) atomicAdd(counts + k, (unsigned long long)s_counts[k]);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
int phb_pin(odw_ctx* ctx, size_t bytes) {
  if (ctx->phb_pin_bytes >= bytes && ctx->phb_pin_p) return ODW_OK;
  if (ctx->phb_pin_p) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->phb_pin_p); }
  ctx->phb_pin_p = nullptr;
  ctx->phb_pin_bytes = 0;
  HIPCHK(ctx, hipHostMalloc(&ctx->phb_pin_p, bytes, hipHostMallocDefault));
  ctx->phb_pin_bytes = bytes;
  return ODW_OK;
}
int phb_event(odw_ctx* ctx) {
  if (!ctx->phb_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->phb_ev, hipEventDisableTiming));
  HIPCHK(ctx, hipEventRecord(ctx->phb_ev, ctx->stream));
  return ODW_OK;
}
// 0: done, 1: not yet (wait = false), < 0: error
int phb_wait(odw_ctx* ctx, bool wait) {
  if (!ctx->phb_ev) return 0;
  if (wait) {
    if (hipEventSynchronize(ctx->phb_ev) != hipSuccess) return -1;
    return 0;
  }
  const hipError_t e = hipEventQuery(ctx->phb_ev);
  if (e == hipSuccess) return 0;
  if (e == hipErrorNotReady) return 1;
  return -1;
}

// layout of the page-locked block: [scenes S][moment parts S x part_stride][counts S x nbins][planes S x 7][origins S x 2,
// strides S][sample rows S x cap] -- the rows last: their size changes from call to call (the plane search's sample, a
// caller's thinned rows) and must not move what an enqueued piece still writes to
struct PhbPinLayout { size_t scenes, rows, part, counts, planes, extra, total; };
PhbPinLayout phb_layout(int S, uint64_t cap, size_t part_stride, uint64_t nbins) {
  PhbPinLayout L;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.scenes = 0;
  L.part = up((size_t)S * sizeof(PhbScene));
  L.counts = L.part + up((size_t)S * part_stride * sizeof(double));
  L.planes = L.counts + up((size_t)S * nbins * sizeof(uint64_t));
  L.extra = L.planes + up((size_t)S * 7 * sizeof(double));
  L.rows = L.extra + up((size_t)S * 2 * sizeof(double) + (size_t)S * sizeof(uint64_t));
  L.total = L.rows + up((size_t)S * cap * sizeof(odw_hit));
  return L;
}

}  // namespace
namespace {
// every buffer of the chain for S scenes of rays_per_scene rays and `slots` slots each (odw_batch_reserve)
int phb_reserve(odw_ctx* ctx, int S, uint64_t rays_per_scene, uint64_t slots) {
  if (rays_per_scene > (1ull << 28)) return ODW_OK;            // (the chain does not serve such launches)
  const uint64_t n_words = (rays_per_scene + 31) / 32, rays_pad = n_words * 32;
  const uint64_t xy_stride = std::min<uint64_t>(slots, rays_pad);
  const size_t part_stride = (size_t)ctx->n_cu * 8 * 10 + 3;
  int rc;
  if ((rc = ensure(ctx, ctx->phb_row_of, (size_t)S * rays_pad * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_words, (size_t)S * 3 * n_words * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_sel, (size_t)S * slots * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_small, (size_t)S * 4 * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_scenes, (size_t)S * sizeof(PhbScene)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_rows, (size_t)S * kPhbRowsRoom * sizeof(odw_hit)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_x, (size_t)S * xy_stride * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_y, (size_t)S * xy_stride * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_part, (size_t)S * part_stride * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_sel_hist, (size_t)S * kPhSelBlocks * 2 * kPhSelBins * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_hist, (size_t)S * 2 * kPhSelBins * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_cand, (size_t)S * 2 * kPhbCand * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_planes, (size_t)S * 7 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_origins, (size_t)S * 2 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_strides, (size_t)S * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_counts, (size_t)S * kPhbBinsRoom * sizeof(uint64_t)))) return rc;
  size_t tmp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_words, ctx->stream));
  if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
  return phb_pin(ctx, phb_layout(S, kPhbRowsRoom, part_stride, kPhbBinsRoom).total);
}

// select for all segments, enqueued (no wait): popcounts, prefix sums, state, rows by rank
int phb_enqueue_select(odw_ctx* ctx, int32_t group, uint64_t limit, uint64_t cap, const char* who) {
  const int S = ctx->batch_traced;
  if (S < 1 || !ctx->batch_rows_ok || !ctx->batch_hits.p || !ctx->batch_seg_slots)
    return fail(ctx, ODW_ERR_INVALID, std::string(who) + ": no batch was traced with hit rows");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->phb_valid = ctx->phb_projected = false;
  ctx->phb_stage = 0;
  const uint64_t n_rays = ctx->batch_rays, ray0 = ctx->batch_first;
  if (n_rays == 0 || n_rays > (1ull << 28)) return fail(ctx, ODW_ERR_UNSUPPORTED, std::string(who) + ": more than 2^28 rays per scene");
  int rc;
  // every group's rows are wanted and the launch noted each row's slot at its ray's place while it recorded it
  // (DeviceOutputs.row_of): the pass over the rows that would fill the table is not needed
  const bool marked = ctx->batch_marked && group < 0;
  const uint64_t n_words = (n_rays + 31) / 32, rays_pad = n_words * 32, slots = ctx->batch_seg_slots;
  if (!marked && (rc = ensure(ctx, ctx->phb_row_of, (size_t)S * rays_pad * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_words, (size_t)S * 3 * n_words * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_sel, (size_t)S * slots * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_small, (size_t)S * 4 * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_scenes, (size_t)S * sizeof(PhbScene)))) return rc;
  if (!marked) {
    HIPCHK(ctx, hipMemsetAsync(ctx->phb_small.p, 0, (size_t)S * 4 * sizeof(uint64_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->phb_row_of.p, 0xff, (size_t)S * rays_pad * sizeof(uint32_t), ctx->stream));       // PH_NO_ROW
    const unsigned kgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((slots + 255) / 256, (uint64_t)ctx->n_cu * 8));
    hipLaunchKernelGGL(phb_mark_kernel, dim3(kgrid, S), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->batch_hits.p, slots,
                       (const unsigned long long*)ctx->batch_hit_count.p, (int)group, ray0, n_rays, rays_pad, (uint32_t*)ctx->phb_row_of.p,
                       (unsigned long long*)ctx->phb_small.p);
  }
  ctx->batch_marked = marked;          // (a table filled for one group is not the launch's any more)
  size_t tmp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_words, ctx->stream));
  if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
  const unsigned rgrid = (unsigned)((n_rays + 256 * kPhbPer - 1) / (256 * kPhbPer));
  hipLaunchKernelGGL(phb_popc_kernel, dim3(rgrid, S), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->phb_row_of.p, n_rays, rays_pad,
                     (uint32_t*)ctx->phb_words.p, n_words);
  for (int s = 0; s < S; ++s) {
    uint32_t* bitmap = (uint32_t*)ctx->phb_words.p + (size_t)s * 3 * n_words;
    HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->sort_tmp.p, tmp_bytes, bitmap + n_words, bitmap + 2 * n_words, (int)n_words, ctx->stream));
  }
  hipLaunchKernelGGL(phb_state_kernel, dim3((S + 63) / 64), dim3(64), 0, ctx->stream, (PhbScene*)ctx->phb_scenes.p, S,
                     (const unsigned long long*)ctx->batch_hit_count.p, (const unsigned long long*)ctx->phb_small.p,
                     (const uint32_t*)ctx->phb_words.p, n_words, n_rays, slots, marked ? 1 : 0, limit, cap);
  hipLaunchKernelGGL(phb_rank_kernel, dim3(rgrid, S), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->phb_words.p, n_words,
                     (const uint32_t*)ctx->phb_row_of.p, n_rays, rays_pad, (uint32_t*)ctx->phb_sel.p, slots);
  HIPCHK(ctx, hipGetLastError());
  ctx->phb_group = group;
  ctx->phb_S = S;
  ctx->phb_part_stride = (size_t)ctx->n_cu * 8 * 10 + 3;
  return ODW_OK;
}

// the sample of every ordered scene into the page-locked block (+ the state), enqueued
int phb_enqueue_sample(odw_ctx* ctx, const uint64_t* strides, uint64_t cap) {
  const int S = ctx->phb_S;
  int rc;
  // (room for the largest sample a chain asks for from the start: growing a device buffer or the page-locked block in the
  //  middle of a sweep waits for every stream of the device)
  if ((rc = ensure(ctx, ctx->phb_rows, (size_t)S * std::max<uint64_t>(cap, kPhbRowsRoom) * sizeof(odw_hit)))) return rc;
  const PhbPinLayout L = phb_layout(S, cap, ctx->phb_part_stride, ctx->phb_nbins);
  if ((rc = phb_pin(ctx, phb_layout(S, std::max<uint64_t>(cap, kPhbRowsRoom), ctx->phb_part_stride, std::max<uint64_t>(ctx->phb_nbins, kPhbBinsRoom)).total))) return rc;
  const uint64_t* d_strides = nullptr;
  if (strides) {
    uint64_t* h = (uint64_t*)((char*)ctx->phb_pin_p + L.extra + (size_t)S * 2 * sizeof(double));
    for (int s = 0; s < S; ++s) h[s] = strides[s];
    if ((rc = ensure(ctx, ctx->phb_strides, (size_t)S * sizeof(uint64_t)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->phb_strides.p, h, (size_t)S * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    d_strides = (const uint64_t*)ctx->phb_strides.p;
  }
  hipLaunchKernelGGL(phb_sample_kernel, dim3((unsigned)((cap * 4 + 255) / 256), S), dim3(256), 0, ctx->stream, (const PhbScene*)ctx->phb_scenes.p,
                     d_strides, (const odw_hit*)ctx->batch_hits.p, ctx->batch_seg_slots, (const uint32_t*)ctx->phb_sel.p, (odw_hit*)ctx->phb_rows.p, cap,
                     (uint64_t)0);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.rows, ctx->phb_rows.p, (size_t)S * cap * sizeof(odw_hit), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.scenes, ctx->phb_scenes.p, (size_t)S * sizeof(PhbScene), hipMemcpyDeviceToHost, ctx->stream));
  ctx->phb_cap = cap;
  return ODW_OK;
}

// the per-scene state the host keeps (after a wait): rows, leaving rows, ordered
void phb_note_state(odw_ctx* ctx) {
  const int S = ctx->phb_S;
  const PhbScene* sc = (const PhbScene*)ctx->phb_pin_p;
  ctx->phb_n.assign((size_t)S, 0);
  ctx->phb_leaving.assign((size_t)S, 0);
  ctx->phb_ordered.assign((size_t)S, 0);
  ctx->phb_used.assign((size_t)S, 0);
  for (int s = 0; s < S; ++s) {
    ctx->phb_n[s] = sc[s].m;
    ctx->phb_leaving[s] = sc[s].leaving;
    ctx->phb_ordered[s] = (int32_t)sc[s].ordered;
    ctx->phb_used[s] = sc[s].used;
  }
  ctx->phb_valid = true;
}

// planes -> projection, extrema, moments, medians (the whole of ph_select_stats), enqueued
int phb_enqueue_project(odw_ctx* ctx, const double* ex, const double* ey, const int32_t* skip) {
  const int S = ctx->phb_S;
  int rc;
  const uint64_t slots = ctx->batch_seg_slots;
  const uint64_t xy_stride = std::min<uint64_t>(slots, (ctx->batch_rays + 31) / 32 * 32);
  const unsigned gmax = (unsigned)ctx->n_cu * 8;
  const size_t part_stride = (size_t)gmax * 10 + 3;
  ctx->phb_xy_stride = xy_stride;
  ctx->phb_part_stride = part_stride;
  if ((rc = ensure(ctx, ctx->phb_x, (size_t)S * xy_stride * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_y, (size_t)S * xy_stride * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_part, (size_t)S * part_stride * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_sel_hist, (size_t)S * kPhSelBlocks * 2 * kPhSelBins * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_hist, (size_t)S * 2 * kPhSelBins * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_cand, (size_t)S * 2 * kPhbCand * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_planes, (size_t)S * 7 * sizeof(double)))) return rc;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, part_stride, ctx->phb_nbins);
  if ((rc = phb_pin(ctx, phb_layout(S, std::max<uint64_t>(ctx->phb_cap, kPhbRowsRoom), part_stride, std::max<uint64_t>(ctx->phb_nbins, kPhbBinsRoom)).total))) return rc;
  double* h = (double*)((char*)ctx->phb_pin_p + L.planes);
  for (int s = 0; s < S; ++s) {
    for (int k = 0; k < 3; ++k) { h[7 * s + k] = ex[3 * s + k]; h[7 * s + 3 + k] = ey[3 * s + k]; }
    h[7 * s + 6] = (skip && skip[s]) ? 1.0 : 0.0;
  }
  PhbScene* scenes = (PhbScene*)ctx->phb_scenes.p;
  const double* X = (const double*)ctx->phb_x.p;
  const double* Y = (const double*)ctx->phb_y.p;
  uint32_t* slices = (uint32_t*)ctx->phb_sel_hist.p;
  uint32_t* hist = (uint32_t*)ctx->phb_hist.p;
  const size_t hist_bytes = (size_t)S * 2 * kPhSelBins * sizeof(uint32_t);
  HIPCHK(ctx, hipMemcpyAsync(ctx->phb_planes.p, h, (size_t)S * 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(phb_planes_kernel, dim3((S + 63) / 64), dim3(64), 0, ctx->stream, scenes, S, (const double*)ctx->phb_planes.p);
  hipLaunchKernelGGL(phb_project_kernel, dim3(gmax, S), dim3(256), 0, ctx->stream, (const PhbScene*)scenes, (const odw_hit*)ctx->batch_hits.p, slots,
                     ctx->batch_pts ? (const double*)ctx->phb_pts.p : (const double*)nullptr, (const uint32_t*)ctx->phb_sel.p, (double*)ctx->phb_x.p, (double*)ctx->phb_y.p, xy_stride, (double*)ctx->phb_part.p, (uint64_t)part_stride);
  hipLaunchKernelGGL(phb_ext_kernel, dim3(S), dim3(256), 0, ctx->stream, scenes, (const double*)ctx->phb_part.p, (uint64_t)part_stride, gmax);
  for (int level = 0; level < 2; ++level) {
    hipLaunchKernelGGL(phb_sel_hist_kernel, dim3(kPhSelBlocks, S), dim3(512), 0, ctx->stream, (const PhbScene*)scenes, X, Y, xy_stride, slices);
    HIPCHK(ctx, hipMemsetAsync(hist, 0, hist_bytes, ctx->stream));
    hipLaunchKernelGGL(phb_sel_sum_kernel, dim3((2 * kPhSelBins + 255) / 256, (kPhSelBlocks + kPhSelSumSlices - 1) / kPhSelSumSlices, S), dim3(256), 0,
                       ctx->stream, (const PhbScene*)scenes, (const uint32_t*)slices, hist);
    if (level == 0) hipLaunchKernelGGL(phb_decide1_kernel, dim3(S), dim3(256), 0, ctx->stream, scenes, (const uint32_t*)hist);
    else hipLaunchKernelGGL(phb_decide2_kernel, dim3(S), dim3(256), 0, ctx->stream, scenes, (const uint32_t*)hist);
  }
  hipLaunchKernelGGL(phb_collect_kernel, dim3(gmax, S), dim3(256), 0, ctx->stream, scenes, X, Y, xy_stride, (double*)ctx->phb_cand.p);
  hipLaunchKernelGGL(phb_pick_kernel, dim3(2, S), dim3(256), 0, ctx->stream, scenes, (const double*)ctx->phb_cand.p);
  HIPCHK(ctx, hipGetLastError());
  return ODW_OK;
}

// state + moment parts to the page-locked block, enqueued
int phb_enqueue_fetch_project(odw_ctx* ctx) {
  const int S = ctx->phb_S;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, ctx->phb_nbins);
  HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.scenes, ctx->phb_scenes.p, (size_t)S * sizeof(PhbScene), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.part, ctx->phb_part.p, (size_t)S * ctx->phb_part_stride * sizeof(double), hipMemcpyDeviceToHost,
                             ctx->stream));
  return ODW_OK;
}

// after the wait: stats [S][8], moments [S][6] from the block (the sums of odw_hits_moments, in its order of additions);
// coordinates the pick kernel could not serve are sorted (sync = true) or reported in flags
int phb_read_project(odw_ctx* ctx, double* stats, double* moments, uint32_t* flags, bool sort_slow) {
  const int S = ctx->phb_S;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, ctx->phb_nbins);
  const PhbScene* sc = (const PhbScene*)((char*)ctx->phb_pin_p + L.scenes);
  const double* part = (const double*)((char*)ctx->phb_pin_p + L.part);
  const unsigned gmax = (unsigned)ctx->n_cu * 8;
  ctx->phb_on.assign((size_t)S, 0);
  int rc;
  for (int s = 0; s < S; ++s) {
    if (flags) flags[s] = sc[s].flags;
    ctx->phb_on[s] = (char)sc[s].on;
    if (!sc[s].on) continue;
    if (sc[s].flags & PHB_BAD_HIST) {
      if (!flags) return fail(ctx, ODW_ERR_DEVICE, "odw_batch_hits_project: histogram of the projection does not add up");
      continue;                  // (reported in flags: the caller measures this scene by itself)
    }
    const uint64_t m = sc[s].m;
    const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)gmax));
    const double* q = part + (size_t)s * ctx->phb_part_stride + (size_t)g * 4;
    double s6[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned b = 0; b < g; ++b)
      for (int k = 0; k < 6; ++k) s6[k] += q[6 * (size_t)b + k];
    for (int k = 0; k < 3; ++k) {
      const double d = s6[k] / (double)m;
      moments[6 * s + k] = q[6 * (size_t)g + k] + d;
      moments[6 * s + 3 + k] = std::max(0.0, s6[3 + k] / (double)m - d * d);
    }
    for (int k = 0; k < 8; ++k) stats[8 * s + k] = sc[s].stats[k];
    for (int a = 0; a < 2; ++a) {
      if (!(sc[s].flags & (a ? PHB_SLOW_Y : PHB_SLOW_X)) || !sort_slow) continue;
      double four[4];              // (a cloud piled up on one value: the sort, for this scene and coordinate alone)
      const double* v = (const double*)(a ? ctx->phb_y.p : ctx->phb_x.p) + (size_t)s * ctx->phb_xy_stride;
      if ((rc = ph_sorted_stats(ctx, v, m, four))) return rc;
      stats[8 * s + 4 * a] = four[0]; stats[8 * s + 4 * a + 1] = four[1];
      if (flags) flags[s] &= ~(a ? PHB_SLOW_Y : PHB_SLOW_X);
    }
  }
  ctx->phb_projected = true;
  return ODW_OK;
}

// the histogram of every projected scene about its origin (origins: the caller's, or null: the device's own medians), enqueued
int phb_enqueue_bin(odw_ctx* ctx, int32_t polar, const double* origins, const double* edges_a, int32_t n_a, const double* edges_b, int32_t n_b) {
  const int S = ctx->phb_S;
  const uint64_t nbins = (uint64_t)(n_a - 1) * (uint64_t)(n_b - 1);
  int rc;
  if ((rc = ensure(ctx, ctx->ph_edges, (size_t)n_a * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->ph_edges_b, (size_t)n_b * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->phb_counts, (size_t)S * std::max<uint64_t>(nbins, kPhbBinsRoom) * sizeof(uint64_t)))) return rc;
  ctx->phb_nbins = nbins;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, nbins);
  if ((rc = phb_pin(ctx, phb_layout(S, std::max<uint64_t>(ctx->phb_cap, kPhbRowsRoom), ctx->phb_part_stride, std::max<uint64_t>(nbins, kPhbBinsRoom)).total))) return rc;
  // (the edges through the page-locked block too: the caller's arrays may go away before the copy runs)
  bool same = ctx->phb_edges_na == n_a && ctx->phb_edges_nb == n_b && ctx->phb_edges_polar == polar && ctx->phb_edges_host.size() == (size_t)(n_a + n_b);
  if (!same) ctx->phb_edges_host.assign((size_t)(n_a + n_b), 0.0);
  for (int k = 0; same && k < n_a; ++k) same = ctx->phb_edges_host[k] == edges_a[k];
  for (int k = 0; same && k < n_b; ++k) same = ctx->phb_edges_host[n_a + k] == edges_b[k];
  if (!same) {
    // (edges change rarely -- a sweep bins every value alike --: a synchronous upload, once)
    for (int k = 0; k < n_a; ++k) ctx->phb_edges_host[k] = edges_a[k];
    for (int k = 0; k < n_b; ++k) ctx->phb_edges_host[n_a + k] = edges_b[k];
    HIPCHK(ctx, hipMemcpyAsync(ctx->ph_edges.p, ctx->phb_edges_host.data(), (size_t)n_a * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ph_edges_b.p, ctx->phb_edges_host.data() + n_a, (size_t)n_b * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PhbBinAccel A;
    ph_make_accel(A, polar, edges_a, n_a, edges_b, n_b);
    if ((rc = ensure(ctx, ctx->phb_accel, sizeof A))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->phb_accel.p, &A, sizeof A, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->phb_edges_na = n_a; ctx->phb_edges_nb = n_b; ctx->phb_edges_polar = polar;
  }
  PhbScene* scenes = (PhbScene*)ctx->phb_scenes.p;
  if (origins) {
    double* h = (double*)((char*)ctx->phb_pin_p + L.extra);
    for (int s = 0; s < 2 * S; ++s) h[s] = origins[s];
    if ((rc = ensure(ctx, ctx->phb_origins, (size_t)S * 2 * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->phb_origins.p, h, (size_t)S * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(phb_origins_kernel, dim3((S + 63) / 64), dim3(64), 0, ctx->stream, scenes, S, (const double*)ctx->phb_origins.p);
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->phb_counts.p, 0, (size_t)S * nbins * sizeof(uint64_t), ctx->stream));
  const unsigned gmax = (unsigned)ctx->n_cu * 8;
  const bool lds_counts = nbins <= (uint64_t)kPhLdsBins;
  const size_t lds = ph_bin_lds_bytes(n_a, n_b, lds_counts ? nbins : 0, polar != 0);
  if (lds_counts)
    hipLaunchKernelGGL((phb_bin_kernel<true>), dim3(gmax, S), dim3(256), lds, ctx->stream, (const PhbScene*)scenes, (const double*)ctx->phb_x.p,
                       (const double*)ctx->phb_y.p, ctx->phb_xy_stride, (int)polar, (const double*)ctx->ph_edges.p, (int)n_a,
                       (const double*)ctx->ph_edges_b.p, (int)n_b, (const PhbBinAccel*)ctx->phb_accel.p, (unsigned long long*)ctx->phb_counts.p);
  else
    hipLaunchKernelGGL((phb_bin_kernel<false>), dim3(gmax, S), dim3(256), lds, ctx->stream, (const PhbScene*)scenes, (const double*)ctx->phb_x.p,
                       (const double*)ctx->phb_y.p, ctx->phb_xy_stride, (int)polar, (const double*)ctx->ph_edges.p, (int)n_a,
                       (const double*)ctx->ph_edges_b.p, (int)n_b, (const PhbBinAccel*)ctx->phb_accel.p, (unsigned long long*)ctx->phb_counts.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.counts, ctx->phb_counts.p, (size_t)S * nbins * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  return ODW_OK;
}

int phb_check_edges(odw_ctx* ctx, const char* who, const double* edges_a, int32_t n_a, const double* edges_b, int32_t n_b) {
  if (!edges_a || !edges_b || n_a < 2 || n_b < 2) return fail(ctx, ODW_ERR_INVALID, std::string(who) + ": bad argument");
  for (int k = 1; k < n_a; ++k) if (!(edges_a[k] >= edges_a[k - 1])) return fail(ctx, ODW_ERR_INVALID, std::string(who) + ": edges must increase monotonically");
  for (int k = 1; k < n_b; ++k) if (!(edges_b[k] >= edges_b[k - 1])) return fail(ctx, ODW_ERR_INVALID, std::string(who) + ": edges must increase monotonically");
  return ODW_OK;
}

}  // namespace

extern "C" {

// ---- one wait per step (v9) ------------------------------------------------------------------------------------------------
int odw_batch_hits_select(odw_ctx* ctx, int32_t group, uint64_t* n_rows, uint64_t* n_leaving, int32_t* ordered) {
  if (!ctx || !n_rows || !n_leaving || !ordered) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_select: bad argument");
  int rc = phb_enqueue_select(ctx, group, 300, 308, "odw_batch_hits_select");
  if (rc) return rc;
  const int S = ctx->phb_S;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, ctx->phb_nbins);
  if ((rc = phb_pin(ctx, phb_layout(S, std::max<uint64_t>(ctx->phb_cap, kPhbRowsRoom), ctx->phb_part_stride, std::max<uint64_t>(ctx->phb_nbins, kPhbBinsRoom)).total))) return rc;
  HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.scenes, ctx->phb_scenes.p, (size_t)S * sizeof(PhbScene), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  phb_note_state(ctx);
  for (int s = 0; s < S; ++s) { n_rows[s] = ctx->phb_n[s]; n_leaving[s] = ctx->phb_leaving[s]; ordered[s] = ctx->phb_ordered[s]; }
  return ODW_OK;
}

// the rows detectPlaneNormal looks at (hits.py:108-113) of every ordered scene: points[::1 + int(n / limit)] and the
// directions of the same rows (all rows enter, or leaving ones are the majority: DeviceHits._sample).  rows: [S][cap].
// strides (optional, per scene): rows [::strides[s]] instead.
int odw_batch_hits_sample(odw_ctx* ctx, uint64_t limit, const uint64_t* strides, odw_hit* rows, uint64_t cap, uint64_t* n_out) {
  if (!ctx || !rows || !n_out || (limit == 0 && !strides) || cap == 0) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_sample: bad argument");
  if (!ctx->phb_valid) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_sample: odw_batch_hits_select first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int S = ctx->phb_S;
  std::vector<uint64_t> st((size_t)S, 1);
  for (int s = 0; s < S; ++s) {
    n_out[s] = 0;
    const uint64_t m = ctx->phb_n[s];
    if (!ctx->phb_ordered[s] || m == 0) continue;
    st[s] = strides ? std::max<uint64_t>(1, strides[s]) : 1 + m / limit;
    const uint64_t count = (m + st[s] - 1) / st[s];
    if (count > cap) return fail(ctx, ODW_ERR_CAPACITY, "odw_batch_hits_sample: output buffer too small");
    n_out[s] = count;
  }
  int rc = phb_enqueue_sample(ctx, st.data(), cap);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const PhbPinLayout L = phb_layout(S, cap, ctx->phb_part_stride, ctx->phb_nbins);
  std::memcpy(rows, (char*)ctx->phb_pin_p + L.rows, (size_t)S * cap * sizeof(odw_hit));
  return ODW_OK;
}

// X = p . ex, Y = p . ey of every ordered scene's selected points, their two middle elements and extrema
// (stats: [S][8] as odw_hits_project), mean and variance of the points (moments: [S][6]).  skip[s] != 0: leave scene s out.
int odw_batch_hits_project(odw_ctx* ctx, const double* ex, const double* ey, const int32_t* skip, double* stats, double* moments) {
  if (!ctx || !ex || !ey || !stats || !moments) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_project: bad argument");
  if (!ctx->phb_valid) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_project: odw_batch_hits_select first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->phb_projected = false;
  int rc = phb_enqueue_project(ctx, ex, ey, skip);
  if (!rc) rc = phb_enqueue_fetch_project(ctx);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return phb_read_project(ctx, stats, moments, nullptr, true);
}

// numpy.histogram2d of every projected scene about its own origin, the same edges for all (counts: [S][(n_a - 1) (n_b - 1)])
int odw_batch_hits_bin(odw_ctx* ctx, int32_t polar, const double* origins, const double* edges_a, int32_t n_a,
                       const double* edges_b, int32_t n_b, uint64_t* counts) {
  if (!ctx || !origins || !counts) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_bin: bad argument");
  int rc = phb_check_edges(ctx, "odw_batch_hits_bin", edges_a, n_a, edges_b, n_b);
  if (rc) return rc;
  if (!ctx->phb_valid || !ctx->phb_projected) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_bin: odw_batch_hits_project first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if ((rc = phb_enqueue_bin(ctx, polar, origins, edges_a, n_a, edges_b, n_b))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const int S = ctx->phb_S;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, ctx->phb_nbins);
  std::memcpy(counts, (char*)ctx->phb_pin_p + L.counts, (size_t)S * ctx->phb_nbins * sizeof(uint64_t));
  return ODW_OK;
}

// ---- the same chain without waits in between (v10) -----------------------------------------------------------------------------
int odw_batch_hits_begin(odw_ctx* ctx, int32_t group, uint64_t limit) {
  if (!ctx || limit == 0 || limit > 4096) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_begin: bad argument");
  const uint64_t cap = limit + 8;
  int rc = phb_enqueue_select(ctx, group, limit, cap, "odw_batch_hits_begin");
  if (!rc) rc = phb_enqueue_sample(ctx, nullptr, cap);
  if (!rc) rc = phb_event(ctx);
  if (rc) return rc;
  ctx->phb_stage = 1;
  return ODW_OK;
}

int odw_batch_hits_sampled(odw_ctx* ctx, int32_t wait, uint64_t* n_rows, uint64_t* n_leaving, int32_t* ordered, odw_hit* rows, uint64_t cap,
                           uint64_t* n_sample) {
  if (!ctx || !n_rows || !n_leaving || !ordered || !rows || !n_sample) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_sampled: bad argument");
  if (ctx->phb_stage != 1) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_sampled: odw_batch_hits_begin first");
  if (cap < ctx->phb_cap) return fail(ctx, ODW_ERR_CAPACITY, "odw_batch_hits_sampled: output buffer too small");
  const int w = phb_wait(ctx, wait != 0);
  if (w < 0) return fail(ctx, ODW_ERR_DEVICE, "odw_batch_hits_sampled: the device reported an error");
  if (w > 0) return ODW_BUSY;
  phb_note_state(ctx);
  const int S = ctx->phb_S;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, ctx->phb_nbins);
  const PhbScene* sc = (const PhbScene*)((char*)ctx->phb_pin_p + L.scenes);
  const odw_hit* src = (const odw_hit*)((char*)ctx->phb_pin_p + L.rows);
  for (int s = 0; s < S; ++s) {
    n_rows[s] = sc[s].m; n_leaving[s] = sc[s].leaving;
    // (a sample cut short by the buffer cannot happen with cap = limit + 8; reported as "not ordered" if it does)
    ordered[s] = (sc[s].flags & PHB_SAMPLE_CUT) ? 0 : (int32_t)sc[s].ordered;
    n_sample[s] = sc[s].sample_n;
    std::memcpy(rows + (size_t)s * cap, src + (size_t)s * ctx->phb_cap, (size_t)sc[s].sample_n * sizeof(odw_hit));
  }
  ctx->phb_stage = 2;
  return ODW_OK;
}

int odw_batch_hits_measure(odw_ctx* ctx, const double* ex, const double* ey, const int32_t* skip, int32_t polar, const double* edges_a,
                           int32_t n_a, const double* edges_b, int32_t n_b, uint64_t keep) {
  if (!ctx || !ex || !ey) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_measure: bad argument");
  int rc = phb_check_edges(ctx, "odw_batch_hits_measure", edges_a, n_a, edges_b, n_b);
  if (rc) return rc;
  if (ctx->phb_stage != 2 || !ctx->phb_valid) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_measure: odw_batch_hits_sampled first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->phb_projected = false;
  ctx->phb_nbins = (uint64_t)(n_a - 1) * (uint64_t)(n_b - 1);
  rc = phb_enqueue_project(ctx, ex, ey, skip);
  if (!rc) rc = phb_enqueue_bin(ctx, polar, nullptr, edges_a, n_a, edges_b, n_b);
  if (!rc) rc = phb_enqueue_fetch_project(ctx);
  ctx->phb_keep = 0;
  if (!rc && keep) {
    // the rows [::max(1, n // keep)] of every ordered scene ride along (what a notebook that traces `keep` rays per value
    // works on): at most 2 keep - 1 rows per scene
    if (keep > (1ull << 20)) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_measure: keep beyond 2^20 rows");
    const uint64_t cap = 2 * keep + 8;
    const int S = ctx->phb_S;
    if ((rc = ensure(ctx, ctx->phb_rows, (size_t)S * std::max<uint64_t>(cap, kPhbRowsRoom) * sizeof(odw_hit)))) return rc;
    if ((rc = phb_pin(ctx, phb_layout(S, std::max<uint64_t>(cap, kPhbRowsRoom), ctx->phb_part_stride, std::max<uint64_t>(ctx->phb_nbins, kPhbBinsRoom)).total))) return rc;
    const PhbPinLayout L = phb_layout(S, cap, ctx->phb_part_stride, ctx->phb_nbins);
    hipLaunchKernelGGL(phb_sample_kernel, dim3((unsigned)((cap * 4 + 255) / 256), S), dim3(256), 0, ctx->stream, (const PhbScene*)ctx->phb_scenes.p,
                       (const uint64_t*)nullptr, (const odw_hit*)ctx->batch_hits.p, ctx->batch_seg_slots, (const uint32_t*)ctx->phb_sel.p,
                       (odw_hit*)ctx->phb_rows.p, cap, keep);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync((char*)ctx->phb_pin_p + L.rows, ctx->phb_rows.p, (size_t)S * cap * sizeof(odw_hit), hipMemcpyDeviceToHost, ctx->stream));
    ctx->phb_keep = keep;
    ctx->phb_cap = cap;
  }
  if (!rc) rc = phb_event(ctx);
  if (rc) return rc;
  ctx->phb_stage = 3;
  return ODW_OK;
}

int odw_batch_hits_measured(odw_ctx* ctx, int32_t wait, double* stats, double* moments, double* origins, uint64_t* counts, uint32_t* flags,
                            odw_hit* keep_rows, uint64_t keep_cap, uint64_t* n_keep) {
  if (!ctx || !stats || !moments || !origins || !counts || !flags) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_measured: bad argument");
  if (ctx->phb_keep && (!keep_rows || !n_keep || keep_cap < ctx->phb_cap))
    return fail(ctx, ODW_ERR_CAPACITY, "odw_batch_hits_measured: room for 2 keep + 8 rows per scene is needed");
  if (ctx->phb_stage != 3) return fail(ctx, ODW_ERR_INVALID, "odw_batch_hits_measured: odw_batch_hits_measure first");
  const int w = phb_wait(ctx, wait != 0);
  if (w < 0) return fail(ctx, ODW_ERR_DEVICE, "odw_batch_hits_measured: the device reported an error");
  if (w > 0) return ODW_BUSY;
  int rc = phb_read_project(ctx, stats, moments, flags, false);
  if (rc) return rc;
  const int S = ctx->phb_S;
  const PhbPinLayout L = phb_layout(S, ctx->phb_cap, ctx->phb_part_stride, ctx->phb_nbins);
  const PhbScene* sc = (const PhbScene*)((char*)ctx->phb_pin_p + L.scenes);
  for (int s = 0; s < S; ++s) { origins[2 * s] = sc[s].origin[0]; origins[2 * s + 1] = sc[s].origin[1]; }
  std::memcpy(counts, (char*)ctx->phb_pin_p + L.counts, (size_t)S * ctx->phb_nbins * sizeof(uint64_t));
  if (ctx->phb_keep) {
    const odw_hit* src = (const odw_hit*)((char*)ctx->phb_pin_p + L.rows);
    for (int s = 0; s < S; ++s) {
      n_keep[s] = 0;
      if (!sc[s].ordered || sc[s].m == 0) continue;
      const uint64_t stride = sc[s].m / ctx->phb_keep ? sc[s].m / ctx->phb_keep : 1;
      const uint64_t count = std::min<uint64_t>((sc[s].m + stride - 1) / stride, ctx->phb_cap);
      n_keep[s] = count;
      std::memcpy(keep_rows + (size_t)s * keep_cap, src + (size_t)s * ctx->phb_cap, (size_t)count * sizeof(odw_hit));
    }
  }
  ctx->phb_stage = 4;
  return ODW_OK;
}

}  // extern "C"
