// odw_posthoc.hip -- the reference's post-hoc detector binning on the hit rows where they are.
//
// Reference: Hits.histogram (jupyter_utils/hits.py:176-193) = detectPlaneNormal on a thinned
// sample (:96-174) -> planeProject3dPoints (:62-94) -> Histogram.__init__ (histogram.py:24-57):
// median origin, numpy.histogram2d of (X, Y) or of (arctan2(X, Y), hypot(X, Y)).  The plane
// search works on <= 300 rows and stays on the host (its arithmetic is numpy's, bit for bit);
// everything that touches all M rows runs here, on the rows in HBM:
//   select   rows of one group in (ray, bounce) order   radix sort of (ray index -> slot) pairs
//   gather   every k-th selected row                     (the thinned sample, <= 300 rows)
//   project  X = p . ex, Y = p . ey                      one 24-byte gather per row, no fma
//   medians  + extrema of X and Y                        radix sort of the 8-byte keys
//   bin      numpy.histogramdd's rule: searchsorted(edges, v, 'right') - 1, the last edge closed;
//            counts in LDS (<= 8192 bins) or by global atomics
// Included by odw_capi.hip (one translation unit).

namespace {

constexpr int kPhLdsBins = 8192;
constexpr int kPhLdsEdges = 2048;

// key = ray index of a selected row, `sentinel` (above every ray index of the list) for everything else;
// counts[0] += selected rows, counts[1] += those that leave.  Grid-stride, one pair of atomics per block: one
// pair per wave serialised 3e5 memory-side atomics of a 1e7-row list on two addresses (1.9 ms; now 0.1).
// The ordered selection without a sort, for lists in which no ray has two selected rows (an absorbing detector:
// every BASELINE config): the row of every ray that has one (plain stores into a table preset to "none": no atomics --
// an atomicOr per row into a bitmap cost 1.06 ms for 1e7 rows, the memory side's rate for 1e7 returning atomics), the
// bitmap of those rays and its words' popcounts by ballot over the table, the rank of a ray = the set bits before it
// (prefix sums of the popcounts); rows written out by rank.  Three passes over 4-byte tables instead of three radix
// passes over key / value pairs; the same list, since ray indices are the sort key.  A ray with two selected rows
// leaves one store of the two standing: fewer rays marked than rows selected -- the caller sees that and sorts.
// counts as in ph_keys_kernel; oob: a ray index outside the launch's range (cannot happen).
#define PH_NO_ROW 0xffffffffu
__global__ __launch_bounds__(256) void ph_mark_kernel(const odw_hit* __restrict__ hits, uint64_t n, int group, uint64_t ray0, uint64_t n_rays,
                                                      uint32_t* __restrict__ row_of, unsigned long long* __restrict__ counts,
                                                      uint32_t* __restrict__ oob) {
  uint32_t n_sel = 0, n_leave = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t tag = hits[i].tag;
    const bool sel = tag != ODW_TAG_UNUSED && (group < 0 || (int)ODW_HIT_GROUP(tag) == group);
    if (sel) {
      const uint64_t r = ODW_HIT_RAY(tag) - ray0;          // (unsigned: an index below ray0 wraps above n_rays)
      if (r >= n_rays) { *oob = 1u; continue; }            // (cannot happen: hit_ray_begin / end bound the indices)
      row_of[r] = (uint32_t)i;
      n_sel += 1u;
      n_leave += ODW_HIT_ENTERING(tag) ? 0u : 1u;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n_sel += __shfl_xor(n_sel, off); n_leave += __shfl_xor(n_leave, off); }
  __shared__ uint32_t s[4][2];
  if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6][0] = n_sel; s[threadIdx.x >> 6][1] = n_leave; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = s[0][0] + s[1][0] + s[2][0] + s[3][0], b = s[0][1] + s[1][1] + s[2][1] + s[3][1];
    if (a) atomicAdd(counts, (unsigned long long)a);
    if (b) atomicAdd(counts + 1, (unsigned long long)b);
  }
}
// bitmap words and their popcounts from the table: a wave looks at 64 consecutive rays, its ballot is two words
// (blocks of 256 threads: n_rays is padded to whole words by the caller's allocation, rays beyond it read as none)
__global__ __launch_bounds__(256) void ph_popc_kernel(const uint32_t* __restrict__ row_of, uint64_t n_rays, uint32_t* __restrict__ bitmap,
                                                      uint32_t* __restrict__ pop) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool has = r < n_rays && row_of[r] != PH_NO_ROW;
  const uint64_t b = __ballot(has);
  const uint32_t lane = threadIdx.x & 63u;
  if ((lane & 31u) == 0 && r < n_rays) {
    const uint32_t word = (uint32_t)(lane ? b >> 32 : b);
    bitmap[r >> 5] = word;
    pop[r >> 5] = (uint32_t)__popc(word);
  }
}
__global__ void ph_rank_kernel(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ before, const uint32_t* __restrict__ row_of,
                               uint64_t n_rays, uint32_t* __restrict__ out, unsigned long long* __restrict__ marked) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) {                                            // rays that have a row: the caller compares with the rows selected
    const uint64_t w = (n_rays - 1) >> 5;
    *marked = (unsigned long long)before[w] + (unsigned long long)__popc(bitmap[w]);
  }
  if (r < n_rays) {
    const uint32_t word = bitmap[r >> 5], bit = 1u << (r & 31u);
    if (word & bit) out[before[r >> 5] + (uint32_t)__popc(word & (bit - 1u))] = row_of[r];
  }
}

// K: uint32_t when ray indices and the sentinel fit 32 bits (the sort then moves half the bytes), else uint64_t
template <class K>
__global__ __launch_bounds__(256) void ph_keys_kernel(const odw_hit* __restrict__ hits, uint64_t n, int group, uint64_t sentinel,
                                                      K* __restrict__ keys, uint32_t* __restrict__ vals,
                                                      unsigned long long* __restrict__ counts) {
  uint32_t n_sel = 0, n_leave = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t tag = hits[i].tag;
    const bool sel = tag != ODW_TAG_UNUSED && (group < 0 || (int)ODW_HIT_GROUP(tag) == group);
    n_sel += sel ? 1u : 0u;
    n_leave += (sel && !ODW_HIT_ENTERING(tag)) ? 1u : 0u;
    keys[i] = (K)(sel ? ODW_HIT_RAY(tag) : sentinel);    // everything else sorts behind every ray
    vals[i] = (uint32_t)i;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n_sel += __shfl_xor(n_sel, off); n_leave += __shfl_xor(n_leave, off); }
  __shared__ uint32_t s[4][2];
  if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6][0] = n_sel; s[threadIdx.x >> 6][1] = n_leave; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = s[0][0] + s[1][0] + s[2][0] + s[3][0], b = s[0][1] + s[1][1] + s[2][1] + s[3][1];
    if (a) atomicAdd(counts, (unsigned long long)a);
    if (b) atomicAdd(counts + 1, (unsigned long long)b);
  }
}

__global__ void ph_entering_flags_kernel(const odw_hit* __restrict__ hits, const uint32_t* __restrict__ sel, uint64_t m,
                                         uint8_t* __restrict__ flags) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) flags[j] = ODW_HIT_ENTERING(hits[sel[j]].tag) ? 1 : 0;
}

// rows sel[j * stride], j < count: four lanes per 64-byte row
__global__ void ph_gather_strided_kernel(const odw_hit* __restrict__ hits, const uint32_t* __restrict__ sel,
                                         uint64_t stride, uint64_t count, odw_hit* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = t >> 2;
  if (row < count) {
    const double2* src = reinterpret_cast<const double2*>(hits + sel[row * stride]);
    reinterpret_cast<double2*>(out + row)[t & 3] = src[t & 3];
  }
}

// the selected rows as columns: points [m][3], directions [m][3], powers [m], isEntering [m] (int64), ray index [m]
// (int64) -- the arrays the reference pickles per (source, object) (results_store.py:405-457)
__global__ void ph_columns_kernel(const odw_hit* __restrict__ hits, const uint32_t* __restrict__ sel, uint64_t m,
                                  double* __restrict__ points, double* __restrict__ dirs, double* __restrict__ powers,
                                  long long* __restrict__ entering, long long* __restrict__ ray) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const odw_hit h = hits[sel[j]];
  if (points) { points[3 * j] = h.point[0]; points[3 * j + 1] = h.point[1]; points[3 * j + 2] = h.point[2]; }
  if (dirs) { dirs[3 * j] = h.direction[0]; dirs[3 * j + 1] = h.direction[1]; dirs[3 * j + 2] = h.direction[2]; }
  if (powers) powers[j] = h.power;
  if (entering) entering[j] = ODW_HIT_ENTERING(h.tag) ? 1 : 0;
  if (ray) ray[j] = (long long)ODW_HIT_RAY(h.tag);
}

// numpy.dot(points, axis): three products, summed left to right, no contraction; per block the extrema of
// both coordinates (part[4 b ..]: min X, max X, min Y, max Y)
// sums of d and d^2, d = point - the first selected row's point (a provisional centre inside the cloud: the
// variance is E[d^2] - E[d]^2 without the cancellation of raw second moments): one row's share, and a block's sums
// written out -- shared by ph_moment_kernel and by ph_project_kernel, which reads the same points anyway (same grid,
// same order of additions: the same bits from either)
__device__ __forceinline__ void ph_moment_add(double (&s)[6], double px, double py, double pz, double cx, double cy, double cz) {
  const double dx = px - cx, dy = py - cy, dz = pz - cz;
  s[0] += dx; s[1] += dy; s[2] += dz;
  s[3] += dx * dx; s[4] += dy * dy; s[5] += dz * dz;
}
__device__ __forceinline__ void ph_moment_write(double (&s)[6], double cx, double cy, double cz, double* __restrict__ part,
                                                unsigned n_blocks = 0) {       // (0: the launch's grid)
  if (n_blocks == 0) n_blocks = gridDim.x;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += __shfl_xor(s[k], off);
  __shared__ double sh[4][6];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 6; ++k) sh[w][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 6) part[6 * blockIdx.x + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 6) { part[6 * n_blocks] = cx; part[6 * n_blocks + 1] = cy; part[6 * n_blocks + 2] = cz; }
}

// part: [gridDim.x][4] extrema of X, Y | (key 0: the points' moment sums) [gridDim.x][6] + the centre [3]
__global__ __launch_bounds__(256) void ph_project_kernel(const odw_hit* __restrict__ hits, const uint32_t* __restrict__ sel,
                                                         uint64_t m, int key, double ex0, double ex1, double ex2, double ey0,
                                                         double ey1, double ey2, double* __restrict__ X,
                                                         double* __restrict__ Y, double* __restrict__ part) {
#pragma clang fp contract(off)
  double lo_x = INFINITY, hi_x = -INFINITY, lo_y = INFINITY, hi_y = -INFINITY;
  const double* c0 = hits[sel[0]].point;
  const double cx = c0[0], cy = c0[1], cz = c0[2];
  double mom[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    const double* p = key ? hits[sel[j]].direction : hits[sel[j]].point;
    const double a = p[0], b = p[1], c = p[2];
    const double x = a * ex0 + b * ex1 + c * ex2, y = a * ey0 + b * ey1 + c * ey2;
    X[j] = x;
    Y[j] = y;
    lo_x = fmin(lo_x, x); hi_x = fmax(hi_x, x);
    lo_y = fmin(lo_y, y); hi_y = fmax(hi_y, y);
    if (key == 0) ph_moment_add(mom, a, b, c, cx, cy, cz);
  }
  if (key == 0) ph_moment_write(mom, cx, cy, cz, part + 4 * (size_t)gridDim.x);
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

// Medians by selection instead of two full radix sorts (numpy.median needs the one or two middle elements only):
// a histogram of kPhSelBins equal bins over [lo, hi] per coordinate -> the host finds the bins that hold the middle
// ranks -> a second histogram of the elements of those bins, kPhSelBins times finer -> the elements of the fine
// bins that hold the ranks are collected and the host picks the ranks among them.  The bin of a value is a
// monotone function of it, so every element of a lower bin is <= every element of a higher one: exact.
// Counting happens in LDS (a focused spot piles 1e7 values into a few hundred bins: global atomics on them
// took 1.8 ms), every block writes its counts to a slice of its own, a second kernel adds the slices.
constexpr int kPhSelBins = 4096;
constexpr int kPhSelBlocks = 512;
struct PhSel {
  double lo[2], sc[2];        // coarse bins of X, Y
  uint32_t c_lo[2], c_hi[2];  // pass 2 / collect: only elements whose coarse bin lies in [c_lo, c_hi]
  double flo[2], fsc[2];      // fine bins of those elements
  uint32_t f0[2], f1[2];      // collect: fine bins wanted
  int fine;                   // 0: first pass
};
__device__ __forceinline__ uint32_t ph_sel_bin(double v, double lo, double scale) {
  const double f = (v - lo) * scale;                       // (NaN: bin 0 -- NaN coordinates do not occur in hit rows)
  return f >= (double)(kPhSelBins - 1) ? (uint32_t)(kPhSelBins - 1) : (f > 0 ? (uint32_t)f : 0u);
}
__global__ __launch_bounds__(512) void ph_sel_hist_kernel(const double* __restrict__ X, const double* __restrict__ Y, uint64_t m,
                                                          const PhSel P, uint32_t* __restrict__ slices) {
  __shared__ uint32_t h[2 * kPhSelBins];
  for (int k = threadIdx.x; k < 2 * kPhSelBins; k += blockDim.x) h[k] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
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
  uint32_t* out = slices + (size_t)blockIdx.x * 2 * kPhSelBins;
  for (int k = threadIdx.x; k < 2 * kPhSelBins; k += blockDim.x) out[k] = h[k];
}
// hist (zeroed by the caller) += the slices: block (x, y) sums 32 slices for 256 bins (one block per 256 bins alone
// read the 16 MB of slices with 32 of the 256 CUs: 75 us)
constexpr int kPhSelSumSlices = 32;
__global__ __launch_bounds__(256) void ph_sel_sum_kernel(const uint32_t* __restrict__ slices, int n_slices, uint32_t* __restrict__ hist) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < 2 * kPhSelBins) {
    const int b0 = blockIdx.y * kPhSelSumSlices, b1 = min(n_slices, b0 + kPhSelSumSlices);
    uint32_t s = 0;
    for (int b = b0; b < b1; ++b) s += slices[(size_t)b * 2 * kPhSelBins + k];
    if (s) atomicAdd(hist + k, s);
  }
}
// elements of X / Y whose coarse bin lies in [c_lo, c_hi] and whose fine bin is f0 or f1, appended to out
// (X: from 0, Y: from off_y); one atomic per wave and coordinate
__global__ __launch_bounds__(256) void ph_sel_collect_kernel(const double* __restrict__ X, const double* __restrict__ Y, uint64_t m,
                                                             const PhSel P, double* __restrict__ out, uint64_t off_y,
                                                             unsigned long long* __restrict__ n_out) {
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
        // (every lane of the wave is here -- the loop bounds are wave-uniform --: lane 0 reserves, readfirstlane
        //  reads lane 0)
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_out + a, (unsigned long long)__popcll(b));
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
        if (want) out[(a ? off_y : 0) + base + __popcll(b & ((1ull << lane) - 1ull))] = v[a];
      }
    }
  }
}

// the two coordinates Histogram bins: cartesian (X - ox, Y - oy); polar (arctan2(X, Y), sqrt(X^2 + Y^2))
__device__ __forceinline__ void ph_coords(double x, double y, double ox, double oy, int polar, double& a, double& b) {
#pragma clang fp contract(off)
  x = x - ox;
  y = y - oy;
  if (polar) {
    a = atan2(x, y);
    b = sqrt(x * x + y * y);
  } else {
    a = x;
    b = y;
  }
}

// min / max of both coordinates (for bin counts given as integers: numpy takes the data's range)
__global__ void ph_range_kernel(const double* __restrict__ X, const double* __restrict__ Y, uint64_t m, double ox,
                                double oy, int polar, double* __restrict__ part) {
  double lo_a = INFINITY, hi_a = -INFINITY, lo_b = INFINITY, hi_b = -INFINITY;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    double a, b;
    ph_coords(X[j], Y[j], ox, oy, polar, a, b);
    lo_a = fmin(lo_a, a); hi_a = fmax(hi_a, a);
    lo_b = fmin(lo_b, b); hi_b = fmax(hi_b, b);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo_a = fmin(lo_a, __shfl_xor(lo_a, off)); hi_a = fmax(hi_a, __shfl_xor(hi_a, off));
    lo_b = fmin(lo_b, __shfl_xor(lo_b, off)); hi_b = fmax(hi_b, __shfl_xor(hi_b, off));
  }
  __shared__ double s[4][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[w][0] = lo_a; s[w][1] = hi_a; s[w][2] = lo_b; s[w][3] = hi_b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      s[0][0] = fmin(s[0][0], s[k][0]); s[0][1] = fmax(s[0][1], s[k][1]);
      s[0][2] = fmin(s[0][2], s[k][2]); s[0][3] = fmax(s[0][3], s[k][3]);
    }
    for (int k = 0; k < 4; ++k) part[4 * blockIdx.x + k] = s[0][k];
  }
}

// numpy.searchsorted(edges, v, side='right') - 1, with v == edges[-1] moved into the last bin
// (numpy/lib/_histograms_impl.py, histogramdd); -1: outside (or NaN)
__device__ __forceinline__ int ph_bin(const double* __restrict__ edges, int n, double v) {
  if (!(v >= edges[0]) || !(v <= edges[n - 1])) return -1;
  int lo = 0, hi = n;                 // first index with edges[i] > v
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (edges[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return v == edges[n - 1] ? n - 2 : lo - 1;
}

// ---- binning without arctan2 and without a search over all edges ------------------------------------------------------------
// The polar histogram of a sweep costs more than the projection: per row an arctan2 (~100 instructions) and a binary search
// over 500 radial edges (9 dependent LDS reads).  Both answers can be had exactly without them:
//  * azimuth (<= kPhbAzEdges edges): whether arctan2(x, y) >= e is the sign of the cross product with the edge's direction
//    (sin e, cos e), once the half (x >= 0: angle >= 0) is known.  Rows within 2e-15 rad of an edge or of the half's border
//    -- where the rounding of arctan2 itself decides -- take arctan2 and the search, as before (none in 1e7 rows, typically);
//  * radius (positive, finite edges): the bit pattern of a positive double grows with its value, so a table over equal
//    steps of the pattern between the first and the last edge (kPhbGuide entries, made by the host) gives the bin of a
//    bucket's first value; the row's bin lies between that and the next entry's -- usually the same.
// The counts are those of ph_bin_kernel (integers; the per-segment entry points keep the plain kernel: the tests hold the
// two against each other).
constexpr int kPhbAzEdges = 8;
constexpr int kPhbGuide = 4096;
struct PhbBinAccel {
  int32_t az_on, n_az;
  double az_s[kPhbAzEdges], az_c[kPhbAzEdges];
  int32_t az_kind[kPhbAzEdges];        // 0: below every angle (e < -pi), 1: an angle, 2: above every angle (e > pi)
  int32_t r_on, r_shift;
  uint64_t r_bits0;
  uint16_t guide[kPhbGuide + 2];
};

// bin of v among the edges [lo, hi] known to bracket it (edges[lo] <= v, bin <= hi): numpy.searchsorted(..., 'right') - 1
__device__ __forceinline__ int phb_bin_between(const double* __restrict__ edges, int lo, int hi, double v) {
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (edges[mid] <= v) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ double ph_minus(double a, double b) { return a - b; }
// one row's pair of bins (ia, ib; -1: outside): numpy.histogram2d's rule, polar rows through the tables above where they apply
__device__ __forceinline__ void ph_bin_pair(const PhbBinAccel* __restrict__ accel, bool az_fast, bool r_fast, const uint16_t* __restrict__ s_guide,
                                            int polar, double x, double y, const double* __restrict__ ea, int na,
                                            const double* __restrict__ eb, int nb, int& ia, int& ib) {
#pragma clang fp contract(off)
  if (polar) {
    const double b = sqrt(x * x + y * y);
    if (r_fast) {
      const double e_b0 = eb[0], e_b1 = eb[nb - 1];
      if (!(b >= e_b0) || !(b <= e_b1)) {
        ib = -1;
      } else if (b == e_b1) {
        ib = nb - 2;
      } else {
        const uint32_t k = (uint32_t)(((uint64_t)__double_as_longlong(b) - accel->r_bits0) >> accel->r_shift);
        ib = phb_bin_between(eb, (int)s_guide[k], (int)s_guide[k + 1], b);
      }
    } else {
      ib = ph_bin(eb, nb, b);
    }
    bool slow = !az_fast;
    ia = -1;
    if (az_fast) {
      const int n_az = accel->n_az;
      const double zone = 2e-15 * (fabs(x) + fabs(y));
      slow = !(fabs(x) > zone);                          // (the border between the halves, NaN, the origin itself)
      const bool upper = x > 0;                          // arctan2(x, y) > 0
      int cnt = 0;
      for (int k = 0; k < n_az; ++k) {
        const int kind = accel->az_kind[k];
        const double s = accel->az_s[k], c = accel->az_c[k];
        const double cross = x * c - y * s;
        slow = slow || (kind == 1 && !(fabs(cross) > zone));
        // e >= 0: the angle reaches it only in the upper half, and there iff the cross product says so;
        // e < 0: every angle of the upper half is above it, in the lower half the cross product decides
        const bool e_nonneg = s > 0 || (s == 0 && c > 0);
        const bool ge = kind == 0 || (kind == 1 && (e_nonneg ? (upper && cross >= 0) : (upper || cross >= 0)));
        cnt += ge ? 1 : 0;
      }
      ia = (cnt == 0 || cnt == n_az) ? -1 : cnt - 1;
    }
    if (slow) ia = ph_bin(ea, na, atan2(x, y));          // (rare: under the wave's divergence)
  } else {
    ia = ph_bin(ea, na, x);
    ib = ph_bin(eb, nb, y);
  }
}

// the tables of PhbBinAccel for a pair of edge arrays (host)
void ph_make_accel(PhbBinAccel& A, int polar, const double* ea, int na, const double* eb, int nb) {
  std::memset(&A, 0, sizeof A);
  if (!polar) return;
  if (na <= kPhbAzEdges && !getenv("ODW_BIN_PLAIN")) {
    bool ok = true;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < na; ++k) {
      const double e = ea[k];
      if (!std::isfinite(e)) { ok = false; break; }
      // (edges within rounding of +-pi: whether an angle reaches them is a matter of arctan2's last bit everywhere on
      //  the negative y axis -- the cross product's zone test sends those rows to arctan2, so they may stay "angles")
      A.az_kind[k] = e < -pi - 1e-12 ? 0 : (e > pi + 1e-12 ? 2 : 1);
      A.az_s[k] = std::sin(e);
      A.az_c[k] = std::cos(e);
    }
    // (the last edge is closed: an angle equal to it counts -- equality lies inside the zone, arctan2 decides)
    A.az_on = ok ? 1 : 0;
    A.n_az = na;
  }
  if (nb >= 2 && nb <= 60000 && !getenv("ODW_BIN_PLAIN")) {
    bool ok = true;
    for (int k = 0; k < nb; ++k) ok = ok && std::isfinite(eb[k]) && eb[k] > 0;
    if (ok && eb[nb - 1] > eb[0]) {
      uint64_t b0, b1;
      std::memcpy(&b0, &eb[0], 8);
      std::memcpy(&b1, &eb[nb - 1], 8);
      int shift = 0;
      while (((b1 - b0) >> shift) >= (uint64_t)kPhbGuide) ++shift;
      A.r_bits0 = b0;
      A.r_shift = shift;
      // guide[k] = bin of the first value of bucket k (searchsorted(edges, v, 'right') - 1); the last entries = the last bin
      int at = 0;
      for (int k = 0; k <= kPhbGuide + 1; ++k) {
        const uint64_t bits = b0 + ((uint64_t)k << shift);
        double v;
        std::memcpy(&v, &bits, 8);
        if (bits > b1) v = eb[nb - 1];
        while (at + 1 < nb && eb[at + 1] <= v) ++at;
        A.guide[k] = (uint16_t)std::min(at, nb - 2);
      }
      A.r_on = 1;
    }
  }
}

// dynamic LDS of the bin kernels: edges (if they fit), counts (if kept in LDS), the radius guide (polar)
size_t ph_bin_lds_bytes(int na, int nb, uint64_t lds_bins, bool polar) {
  size_t bytes = (na + nb <= kPhLdsEdges) ? (size_t)(na + nb) * sizeof(double) : 0;
  bytes += (size_t)lds_bins * sizeof(uint32_t);
  if (polar) bytes += (size_t)(kPhbGuide + 2) * sizeof(uint16_t);
  return bytes + 16;
}

template <bool LDS_COUNTS>
__global__ __launch_bounds__(256) void ph_bin_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                                                     uint64_t m, double ox, double oy, int polar,
                                                     const double* __restrict__ edges_a, int na,
                                                     const double* __restrict__ edges_b, int nb,
                                                     const PhbBinAccel* __restrict__ accel, unsigned long long* __restrict__ counts) {
  extern __shared__ double ph_bin_lds[];          // [edges | counts | guide], sized by the caller (ph_bin_lds_bytes)
  const bool edges_in_lds = na + nb <= kPhLdsEdges;
  const int nbins = (na - 1) * (nb - 1);
  double* s_edges = ph_bin_lds;
  uint32_t* s_counts = reinterpret_cast<uint32_t*>(ph_bin_lds + (edges_in_lds ? na + nb : 0));
  uint16_t* s_guide = reinterpret_cast<uint16_t*>(s_counts + (LDS_COUNTS ? nbins : 0));
  const bool az_fast = polar && accel && accel->az_on && edges_in_lds;
  const bool r_fast = polar && accel && accel->r_on && edges_in_lds;
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
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    int ia, ib;
    if (accel) {
      ph_bin_pair(accel, az_fast, r_fast, s_guide, polar, ph_minus(X[j], ox), ph_minus(Y[j], oy), ea, na, eb, nb, ia, ib);
    } else {
      double a, b;
      ph_coords(X[j], Y[j], ox, oy, polar, a, b);
      ia = ph_bin(ea, na, a);
      ib = ph_bin(eb, nb, b);
    }
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

// the moment sums by themselves (ph_moment_add / ph_moment_write above), one pass over the rows
__global__ __launch_bounds__(256) void ph_moment_kernel(const odw_hit* __restrict__ hits, const uint32_t* __restrict__ sel, uint64_t m,
                                                        double* __restrict__ part) {
  const double* c = hits[sel[0]].point;
  const double cx = c[0], cy = c[1], cz = c[2];
  double s[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    const double* p = hits[sel[j]].point;
    ph_moment_add(s, p[0], p[1], p[2], cx, cy, cz);
  }
  ph_moment_write(s, cx, cy, cz, part);
}

int ph_need_selection(odw_ctx* ctx, const char* who) {
  if (!ctx->ph_valid) return fail(ctx, ODW_ERR_INVALID, std::string(who) + ": odw_hits_select first (the hit list changed since)");
  return ODW_OK;
}

int ph_need_projection(odw_ctx* ctx, const char* who) {
  int rc = ph_need_selection(ctx, who);
  if (rc) return rc;
  if (!ctx->ph_projected) return fail(ctx, ODW_ERR_INVALID, std::string(who) + ": odw_hits_project first");
  return ODW_OK;
}

// sorted copy of v[0..m) -> the two middle values and the extrema (the fallback of ph_select_stats)
int ph_sorted_stats(odw_ctx* ctx, const double* v, uint64_t m, double out[4]) {
  int rc;
  if ((rc = ensure(ctx, ctx->ph_sorted, m * sizeof(double)))) return rc;
  size_t tmp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, v, (double*)ctx->ph_sorted.p, (int)m, 0, 64, ctx->stream));
  if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(ctx->sort_tmp.p, tmp_bytes, v, (double*)ctx->ph_sorted.p, (int)m, 0, 64,
                                                ctx->stream));
  const double* s = (const double*)ctx->ph_sorted.p;
  const uint64_t at[4] = {(m - 1) / 2, m / 2, 0, m - 1};      // numpy.median: mean of these two
  for (int k = 0; k < 4; ++k)
    HIPCHK(ctx, hipMemcpyAsync(out + k, s + at[k], sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

// The two middle elements (numpy.median) of X and of Y, given their extrema: two histogram passes, the elements of
// the fine bins that hold the middle ranks, the ranks among them.  stats = {x mid lo, x mid hi, min x, max x, y ...}.
// A coordinate whose fine bins still hold more than kPhSelMax elements (a cloud piled up on one value) takes the sort.
constexpr uint64_t kPhSelMax = 1u << 21;
// bins that hold ranks k_lo <= k_hi in a histogram: bin[2], elements below each
static bool ph_rank_bins(const uint32_t* h, uint64_t k_lo, uint64_t k_hi, uint32_t bin[2], uint64_t below[2]) {
  uint64_t run = 0;
  int found = 0;
  for (int b = 0; b < kPhSelBins && found < 2; ++b) {
    const uint64_t next = run + h[b];
    while (found < 2 && (found == 0 ? k_lo : k_hi) < next) { bin[found] = (uint32_t)b; below[found] = run; ++found; }
    run = next;
  }
  return found == 2;
}
int ph_select_stats(odw_ctx* ctx, uint64_t m, const double ext[4], double stats[8]) {
  int rc;
  const double* X = (const double*)ctx->ph_x.p;
  const double* Y = (const double*)ctx->ph_y.p;
  stats[2] = ext[0]; stats[3] = ext[1]; stats[6] = ext[2]; stats[7] = ext[3];
  const uint64_t k_lo = (m - 1) / 2, k_hi = m / 2;
  PhSel P;
  std::memset(&P, 0, sizeof P);
  double width[2];
  for (int a = 0; a < 2; ++a) {
    P.lo[a] = ext[2 * a];
    width[a] = ext[2 * a + 1] - ext[2 * a];
    P.sc[a] = (width[a] > 0 && width[a] < INFINITY) ? (double)kPhSelBins / width[a] : 0.0;
  }
  const size_t slice_bytes = (size_t)kPhSelBlocks * 2 * kPhSelBins * sizeof(uint32_t);
  const size_t hist_bytes = 2 * (size_t)kPhSelBins * sizeof(uint32_t);
  if ((rc = ensure(ctx, ctx->ph_sel_hist, slice_bytes + hist_bytes + 64))) return rc;
  uint32_t* slices = (uint32_t*)ctx->ph_sel_hist.p;
  uint32_t* d_hist = (uint32_t*)((char*)ctx->ph_sel_hist.p + slice_bytes);
  unsigned long long* d_count = (unsigned long long*)((char*)ctx->ph_sel_hist.p + slice_bytes + hist_bytes);
  const unsigned hgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 511) / 512, (uint64_t)kPhSelBlocks));
  std::vector<uint32_t> hist(2 * (size_t)kPhSelBins);
  auto histogram = [&]() -> int {
    hipLaunchKernelGGL(ph_sel_hist_kernel, dim3(hgrid), dim3(512), 0, ctx->stream, X, Y, m, P, slices);
    HIPCHK(ctx, hipMemsetAsync(d_hist, 0, hist_bytes, ctx->stream));
    hipLaunchKernelGGL(ph_sel_sum_kernel, dim3((2 * kPhSelBins + 255) / 256, (hgrid + kPhSelSumSlices - 1) / kPhSelSumSlices),
                       dim3(256), 0, ctx->stream, (const uint32_t*)slices, (int)hgrid, d_hist);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(hist.data(), d_hist, hist_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ODW_OK;
  };
  if ((rc = histogram())) return rc;
  uint32_t cbin[2][2], fbin[2][2];
  uint64_t cbelow[2][2], fbelow[2][2], take[2] = {0, 0};
  bool by_sort[2] = {false, false};
  for (int a = 0; a < 2; ++a) {
    if (!ph_rank_bins(hist.data() + (size_t)a * kPhSelBins, k_lo, k_hi, cbin[a], cbelow[a]))
      return fail(ctx, ODW_ERR_DEVICE, "odw_hits_project: histogram of the projection does not add up");
    P.c_lo[a] = cbin[a][0]; P.c_hi[a] = cbin[a][1];
    // fine bins over the value range of the coarse bins [c_lo, c_hi]
    const double w = width[a] / (double)kPhSelBins;
    P.flo[a] = P.lo[a] + (double)P.c_lo[a] * w;
    const double fw = (double)(P.c_hi[a] - P.c_lo[a] + 1) * w;
    P.fsc[a] = (fw > 0 && fw < INFINITY) ? (double)kPhSelBins / fw : 0.0;
  }
  P.fine = 1;
  if ((rc = histogram())) return rc;
  for (int a = 0; a < 2; ++a) {
    // ranks among the elements of the coarse bins [c_lo, c_hi]: everything below c_lo lies below
    if (!ph_rank_bins(hist.data() + (size_t)a * kPhSelBins, k_lo - cbelow[a][0], k_hi - cbelow[a][0], fbin[a], fbelow[a]))
      return fail(ctx, ODW_ERR_DEVICE, "odw_hits_project: second histogram of the projection does not add up");
    P.f0[a] = fbin[a][0]; P.f1[a] = fbin[a][1];
    const uint32_t* h = hist.data() + (size_t)a * kPhSelBins;
    take[a] = h[fbin[a][0]] + (fbin[a][1] != fbin[a][0] ? h[fbin[a][1]] : 0u);
    by_sort[a] = take[a] > kPhSelMax;
    if (by_sort[a]) { P.c_lo[a] = 1; P.c_hi[a] = 0; }       // (collects nothing)
  }
  const uint64_t off_y = by_sort[0] ? 0 : take[0];
  const uint64_t n_out = (by_sort[0] ? 0 : take[0]) + (by_sort[1] ? 0 : take[1]);
  std::vector<double> cand(n_out);
  if (n_out) {
    if ((rc = ensure(ctx, ctx->ph_sorted, n_out * sizeof(double)))) return rc;
    HIPCHK(ctx, hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned long long), ctx->stream));
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)ctx->n_cu * 8));
    hipLaunchKernelGGL(ph_sel_collect_kernel, dim3(grid), dim3(256), 0, ctx->stream, X, Y, m, P, (double*)ctx->ph_sorted.p, off_y,
                       d_count);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(cand.data(), ctx->ph_sorted.p, n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  for (int a = 0; a < 2; ++a) {
    if (by_sort[a]) {
      double four[4];
      if ((rc = ph_sorted_stats(ctx, a ? Y : X, m, four))) return rc;
      stats[4 * a] = four[0]; stats[4 * a + 1] = four[1];
      continue;
    }
    double* c = cand.data() + (a ? off_y : 0);
    std::sort(c, c + take[a]);
    // the collected elements are those of fine bin f0 (and, if different, f1), in value order after the sort
    const uint64_t r_lo = k_lo - cbelow[a][0], r_hi = k_hi - cbelow[a][0];
    const uint64_t n0 = hist[(size_t)a * kPhSelBins + fbin[a][0]];
    stats[4 * a] = c[r_lo - fbelow[a][0]];
    stats[4 * a + 1] = fbin[a][1] == fbin[a][0] ? c[r_hi - fbelow[a][0]] : c[n0 + (r_hi - fbelow[a][1])];
  }
  return ODW_OK;
}

}  // namespace

extern "C" {

int odw_load_hits(odw_ctx* ctx, const odw_hit* rows, uint64_t n) {
  if (!ctx || (!rows && n)) return fail(ctx, ODW_ERR_INVALID, "odw_load_hits: bad argument");
  int rc = odw_reserve_hits(ctx, std::max<uint64_t>(n, 16));
  if (rc) return rc;
  ctx->ph_valid = false;
  ctx->hit_ray_end = 1ull << 48;     // rows of unknown origin: every bit of the ray index counts
  if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->hits.p, rows, n * sizeof(odw_hit), hipMemcpyHostToDevice, ctx->stream));
  const uint64_t count[2] = {n, 0};
  HIPCHK(ctx, hipMemcpyAsync(ctx->hit_count.p, count, sizeof count, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // the caller's rows and `count` may go away
  return ODW_OK;
}

int odw_hits_select(odw_ctx* ctx, int32_t group, uint64_t* n_rows, uint64_t* n_leaving) {
  if (!ctx || !n_rows) return fail(ctx, ODW_ERR_INVALID, "odw_hits_select: bad argument");
  ctx->ph_valid = ctx->ph_projected = false;
  ctx->ph_moment_grid = 0;           // (the moment sums belong to a selection)
  uint64_t used = 0, have = 0;
  int rc = hit_slots_used(ctx, &used, &have);
  if (rc) return rc;
  ctx->ph_n = ctx->ph_n_entering = 0;
  *n_rows = 0;
  if (n_leaving) *n_leaving = 0;
  if (used > 0x7FFFFFFFull) return fail(ctx, ODW_ERR_CAPACITY, "odw_hits_select: more than 2^31 rows");
  if (used) {
    for (int k = 0; k < 2; ++k) {
      if ((rc = ensure(ctx, ctx->sort_keys[k], used * sizeof(uint64_t)))) return rc;
      if ((rc = ensure(ctx, ctx->sort_vals[k], used * sizeof(uint32_t)))) return rc;
    }
    if ((rc = ensure(ctx, ctx->ph_small, 64))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->ph_small.p, 0, 2 * sizeof(uint64_t), ctx->stream));
    uint64_t* k_in = (uint64_t*)ctx->sort_keys[0].p;
    uint64_t* k_out = (uint64_t*)ctx->sort_keys[1].p;
    uint32_t* v_in = (uint32_t*)ctx->sort_vals[0].p;
    uint32_t* v_out = (uint32_t*)ctx->sort_vals[1].p;
    // ray indices of the list lie below hit_ray_end (launch_trace keeps it): the sort needs their bits only; the
    // sentinel of the rows that are not selected is the largest number of that many bits when no ray has it
    // (1e7 rays: 24 bits = 3 passes of 8 instead of 7; a 25th bit for the sentinel alone made a fourth pass in which
    // every key had the same digit -- the slowest of all, 0.29 ms of 0.86)
    int bits = 48;
    if (ctx->hit_ray_end && ctx->hit_ray_end < (1ull << 48)) { bits = 1; while ((1ull << bits) < ctx->hit_ray_end) ++bits; }
    const bool spare = ctx->hit_ray_end && ctx->hit_ray_end < (1ull << bits);      // ray indices stay below 2^bits - 1
    const uint64_t sentinel = spare ? (1ull << bits) - 1 : 1ull << bits;
    if (spare) --bits;                                                             // (the sorts below take bits + 1)
    const unsigned kgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((used + 255) / 256, (uint64_t)ctx->n_cu * 8));
    size_t tmp_bytes = 0;
    // first the route without a sort (ph_mark_kernel): it gives up when a ray has two selected rows
    static const bool sort_only = getenv("ODW_SELECT_SORT") != nullptr;        // (A/B runs)
    const bool bounded = ctx->hit_ray_end && ctx->hit_ray_end < (1ull << 48) && ctx->hit_ray_begin < ctx->hit_ray_end;
    const uint64_t ray0 = bounded ? ctx->hit_ray_begin : 0, n_rays = bounded ? ctx->hit_ray_end - ray0 : 0;
    if (!sort_only && n_rays && n_rays <= (1ull << 28)) {
      const uint64_t n_words = (n_rays + 31) / 32;
      if ((rc = ensure(ctx, ctx->ph_bitmap, n_words * sizeof(uint32_t)))) return rc;
      if ((rc = ensure(ctx, ctx->ph_before, n_words * 2 * sizeof(uint32_t)))) return rc;       // popcounts | prefix sums
      if ((rc = ensure(ctx, ctx->ph_row_of, n_rays * sizeof(uint32_t)))) return rc;
      uint32_t* bitmap = (uint32_t*)ctx->ph_bitmap.p;
      uint32_t* pop = (uint32_t*)ctx->ph_before.p;
      uint32_t* before = pop + n_words;
      uint32_t* row_of = (uint32_t*)ctx->ph_row_of.p;
      unsigned long long* small = (unsigned long long*)ctx->ph_small.p;             // selected, leaving | rays marked | out-of-range flag
      HIPCHK(ctx, hipMemsetAsync(small + 2, 0, 2 * sizeof(uint64_t), ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(row_of, 0xff, n_rays * sizeof(uint32_t), ctx->stream));       // PH_NO_ROW
      hipLaunchKernelGGL(ph_mark_kernel, dim3(kgrid), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p, used, (int)group,
                         ray0, n_rays, row_of, small, (uint32_t*)(small + 3));
      hipLaunchKernelGGL(ph_popc_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)row_of,
                         n_rays, bitmap, pop);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, pop, before, (int)n_words, ctx->stream));
      if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
      HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->sort_tmp.p, tmp_bytes, pop, before, (int)n_words, ctx->stream));
      hipLaunchKernelGGL(ph_rank_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, ctx->stream, bitmap, before,
                         (const uint32_t*)row_of, n_rays, v_out, small + 2);
      HIPCHK(ctx, hipGetLastError());
      uint64_t c[4] = {0, 0, 0, 0};
      HIPCHK(ctx, hipMemcpyAsync(c, small, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      const bool h_dup = c[3] != 0 || c[2] != c[0];                               // (two rows of one ray: one store of the two stands)
      if (!h_dup) {
        ctx->ph_n = c[0];
        ctx->ph_n_entering = c[0] - c[1];
        ctx->ph_entering_built = false;
        *n_rows = c[0];
        if (n_leaving) *n_leaving = c[1];
        ctx->ph_group = group;
        ctx->ph_valid = true;
        return ODW_OK;
      }
      HIPCHK(ctx, hipMemsetAsync(ctx->ph_small.p, 0, 2 * sizeof(uint64_t), ctx->stream));      // the counts again, with the sort
    }
    static const bool keys64 = getenv("ODW_SELECT_KEYS64") != nullptr;        // (A/B runs)
    if (bits + 1 <= 32 && !keys64) {
      uint32_t* k32_in = (uint32_t*)k_in;
      uint32_t* k32_out = (uint32_t*)k_out;
      hipLaunchKernelGGL(ph_keys_kernel<uint32_t>, dim3(kgrid), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p, used,
                         (int)group, sentinel, k32_in, v_in, (unsigned long long*)ctx->ph_small.p);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k32_in, k32_out, v_in, v_out, (int)used, 0, bits + 1,
                                                     ctx->stream));
      if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k32_in, k32_out, v_in, v_out, (int)used, 0,
                                                     bits + 1, ctx->stream));
    } else {
      hipLaunchKernelGGL(ph_keys_kernel<uint64_t>, dim3(kgrid), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p, used,
                         (int)group, sentinel, k_in, v_in, (unsigned long long*)ctx->ph_small.p);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (int)used, 0, bits + 1,
                                                     ctx->stream));
      if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (int)used, 0,
                                                     bits + 1, ctx->stream));
    }
    uint64_t c[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(c, ctx->ph_small.p, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ph_n = c[0];
    ctx->ph_n_entering = c[0] - c[1];
    ctx->ph_entering_built = false;
    *n_rows = c[0];
    if (n_leaving) *n_leaving = c[1];
  }
  ctx->ph_group = group;
  ctx->ph_valid = true;
  return ODW_OK;
}

int odw_hits_columns(odw_ctx* ctx, double* points, double* directions, double* powers, int64_t* is_entering,
                     int64_t* ray_index, uint64_t capacity, uint64_t* n) {
  if (!ctx || !n) return fail(ctx, ODW_ERR_INVALID, "odw_hits_columns: bad argument");
  int rc = ph_need_selection(ctx, "odw_hits_columns");
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t m = ctx->ph_n;
  *n = m;
  if (m == 0 || !(points || directions || powers || is_entering || ray_index)) return ODW_OK;
  if (m > capacity) return fail(ctx, ODW_ERR_CAPACITY, "odw_hits_columns: output arrays too small");
  // Page-locked destinations (odw_host_alloc: the run loop's arrays) are written by the kernel itself, across PCIe:
  // no staging buffer, no copy commands -- five copies of 30 - 100 MB through one copy engine moved 30 GB/s, the kernel's
  // stores fill the link (ODW_COLUMNS_DIRECT=0: the staged route, also taken for pageable destinations)
  {
    static const bool direct_off = [] { const char* e = getenv("ODW_COLUMNS_DIRECT"); return e && e[0] == '0'; }();
    bool direct = !direct_off;
    void* dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    void* host[5] = {points, directions, powers, is_entering, ray_index};
    for (int k = 0; k < 5 && direct; ++k) {
      if (!host[k]) continue;
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, host[k]) != hipSuccess) { (void)hipGetLastError(); direct = false; break; }
      if (attr.type != hipMemoryTypeHost || !attr.devicePointer) { direct = false; break; }
      dev[k] = attr.devicePointer;
    }
    if (direct) {
      hipLaunchKernelGGL(ph_columns_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream,
                         (const odw_hit*)ctx->hits.p, (const uint32_t*)ctx->sort_vals[1].p, m, (double*)dev[0], (double*)dev[1], (double*)dev[2],
                         (long long*)dev[3], (long long*)dev[4]);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      return ODW_OK;
    }
  }
  // staging: 9 doubles per row (3 + 3 + 1 + 1 + 1), one buffer
  if ((rc = ensure(ctx, ctx->sorted_rows, m * 9 * sizeof(double)))) return rc;
  double* base = (double*)ctx->sorted_rows.p;
  double* d_points = points ? base : nullptr;
  double* d_dirs = directions ? base + 3 * m : nullptr;
  double* d_powers = powers ? base + 6 * m : nullptr;
  long long* d_ent = is_entering ? (long long*)(base + 7 * m) : nullptr;
  long long* d_ray = ray_index ? (long long*)(base + 8 * m) : nullptr;
  hipLaunchKernelGGL(ph_columns_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const odw_hit*)ctx->hits.p, (const uint32_t*)ctx->sort_vals[1].p, m, d_points, d_dirs, d_powers, d_ent,
                     d_ray);
  HIPCHK(ctx, hipGetLastError());
  if (points) HIPCHK(ctx, hipMemcpyAsync(points, d_points, m * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (directions) HIPCHK(ctx, hipMemcpyAsync(directions, d_dirs, m * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (powers) HIPCHK(ctx, hipMemcpyAsync(powers, d_powers, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (is_entering) HIPCHK(ctx, hipMemcpyAsync(is_entering, d_ent, m * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (ray_index) HIPCHK(ctx, hipMemcpyAsync(ray_index, d_ray, m * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

int odw_hits_gather(odw_ctx* ctx, int32_t entering_only, uint64_t stride, odw_hit* out, uint64_t capacity, uint64_t* n) {
  if (!ctx || !n || stride == 0) return fail(ctx, ODW_ERR_INVALID, "odw_hits_gather: bad argument");
  int rc = ph_need_selection(ctx, "odw_hits_gather");
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t* sel = (const uint32_t*)ctx->sort_vals[1].p;
  uint64_t m = ctx->ph_n;
  if (entering_only && ctx->ph_n_entering == ctx->ph_n) {
    // (every selected row enters -- an absorbing detector: the list is the selection itself)
  } else if (entering_only) {
    // rows with isEntering != 0, still in (ray, bounce) order: flags + stream compaction
    if (!ctx->ph_entering_built && m) {
      if ((rc = ensure(ctx, ctx->ph_flags, m))) return rc;
      if ((rc = ensure(ctx, ctx->ph_sel_entering, m * sizeof(uint32_t)))) return rc;
      hipLaunchKernelGGL(ph_entering_flags_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream,
                         (const odw_hit*)ctx->hits.p, sel, m, (uint8_t*)ctx->ph_flags.p);
      HIPCHK(ctx, hipGetLastError());
      size_t tmp_bytes = 0;
      HIPCHK(ctx, hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, sel, (const uint8_t*)ctx->ph_flags.p,
                                                (uint32_t*)ctx->ph_sel_entering.p, (uint64_t*)ctx->ph_small.p, (int)m,
                                                ctx->stream));
      if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
      HIPCHK(ctx, hipcub::DeviceSelect::Flagged(ctx->sort_tmp.p, tmp_bytes, sel, (const uint8_t*)ctx->ph_flags.p,
                                                (uint32_t*)ctx->ph_sel_entering.p, (uint64_t*)ctx->ph_small.p, (int)m,
                                                ctx->stream));
      ctx->ph_entering_built = true;
    }
    sel = (const uint32_t*)ctx->ph_sel_entering.p;
    m = ctx->ph_n_entering;
  }
  const uint64_t count = (m + stride - 1) / stride;
  *n = count;
  if (!out || count == 0) return ODW_OK;
  if (count > capacity) return fail(ctx, ODW_ERR_CAPACITY, "odw_hits_gather: output buffer too small");
  if ((rc = ensure(ctx, ctx->sorted_rows, count * sizeof(odw_hit)))) return rc;
  hipLaunchKernelGGL(ph_gather_strided_kernel, dim3((unsigned)((count * 4 + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const odw_hit*)ctx->hits.p, sel, stride, count, (odw_hit*)ctx->sorted_rows.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->sorted_rows.p, count * sizeof(odw_hit), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

int odw_hits_project(odw_ctx* ctx, int32_t key, const double* ex, const double* ey, double* stats) {
  if (!ctx || !ex || !ey || !stats) return fail(ctx, ODW_ERR_INVALID, "odw_hits_project: bad argument");
  int rc = ph_need_selection(ctx, "odw_hits_project");
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->ph_projected = false;
  const uint64_t m = ctx->ph_n;
  if (m == 0) return fail(ctx, ODW_ERR_INVALID, "odw_hits_project: no rows selected");
  if ((rc = ensure(ctx, ctx->ph_x, m * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->ph_y, m * sizeof(double)))) return rc;
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)ctx->n_cu * 8));
  // (the projection of the points reads what the moments need: their sums ride along, odw_hits_moments finds them)
  const size_t n_part = (size_t)grid * 4 + (key == 0 ? (size_t)grid * 6 + 3 : 0);
  if ((rc = ensure(ctx, ctx->ph_part, n_part * sizeof(double)))) return rc;
  hipLaunchKernelGGL(ph_project_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p,
                     (const uint32_t*)ctx->sort_vals[1].p, m, (int)key, ex[0], ex[1], ex[2], ey[0], ey[1], ey[2],
                     (double*)ctx->ph_x.p, (double*)ctx->ph_y.p, (double*)ctx->ph_part.p);
  HIPCHK(ctx, hipGetLastError());
  std::vector<double> part(n_part);
  HIPCHK(ctx, hipMemcpyAsync(part.data(), ctx->ph_part.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (key == 0) {
    ctx->ph_moment_sums.assign(part.begin() + (size_t)grid * 4, part.end());
    ctx->ph_moment_grid = grid;
  }
  double ext[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
  for (unsigned b = 0; b < grid; ++b) {
    ext[0] = std::fmin(ext[0], part[4 * b]); ext[1] = std::fmax(ext[1], part[4 * b + 1]);
    ext[2] = std::fmin(ext[2], part[4 * b + 2]); ext[3] = std::fmax(ext[3], part[4 * b + 3]);
  }
  if ((rc = ph_select_stats(ctx, m, ext, stats))) return rc;
  ctx->ph_projected = true;
  return ODW_OK;
}

int odw_hits_range(odw_ctx* ctx, int32_t polar, const double* origin, double* range) {
  if (!ctx || !origin || !range) return fail(ctx, ODW_ERR_INVALID, "odw_hits_range: bad argument");
  int rc = ph_need_projection(ctx, "odw_hits_range");
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t m = ctx->ph_n;
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)ctx->n_cu * 8));
  if ((rc = ensure(ctx, ctx->ph_part, (size_t)grid * 4 * sizeof(double)))) return rc;
  hipLaunchKernelGGL(ph_range_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->ph_x.p,
                     (const double*)ctx->ph_y.p, m, origin[0], origin[1], (int)polar, (double*)ctx->ph_part.p);
  HIPCHK(ctx, hipGetLastError());
  std::vector<double> part((size_t)grid * 4);
  HIPCHK(ctx, hipMemcpyAsync(part.data(), ctx->ph_part.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  range[0] = range[2] = INFINITY;
  range[1] = range[3] = -INFINITY;
  for (unsigned b = 0; b < grid; ++b) {
    range[0] = std::fmin(range[0], part[4 * b]); range[1] = std::fmax(range[1], part[4 * b + 1]);
    range[2] = std::fmin(range[2], part[4 * b + 2]); range[3] = std::fmax(range[3], part[4 * b + 3]);
  }
  return ODW_OK;
}

int odw_hits_bin(odw_ctx* ctx, int32_t polar, const double* origin, const double* edges_a, int32_t n_a,
                 const double* edges_b, int32_t n_b, uint64_t* counts) {
  if (!ctx || !origin || !edges_a || !edges_b || !counts || n_a < 2 || n_b < 2)
    return fail(ctx, ODW_ERR_INVALID, "odw_hits_bin: bad argument");
  for (int k = 1; k < n_a; ++k) if (!(edges_a[k] >= edges_a[k - 1])) return fail(ctx, ODW_ERR_INVALID, "odw_hits_bin: edges must increase monotonically");
  for (int k = 1; k < n_b; ++k) if (!(edges_b[k] >= edges_b[k - 1])) return fail(ctx, ODW_ERR_INVALID, "odw_hits_bin: edges must increase monotonically");
  int rc = ph_need_projection(ctx, "odw_hits_bin");
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t m = ctx->ph_n;
  const uint64_t nbins = (uint64_t)(n_a - 1) * (uint64_t)(n_b - 1);
  if ((rc = upload(ctx, ctx->ph_edges, edges_a, (size_t)n_a * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->ph_edges_b, (size_t)n_b * sizeof(double)))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->ph_edges_b.p, edges_b, (size_t)n_b * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if ((rc = ensure(ctx, ctx->ph_counts, nbins * sizeof(uint64_t)))) return rc;
  HIPCHK(ctx, hipMemsetAsync(ctx->ph_counts.p, 0, nbins * sizeof(uint64_t), ctx->stream));
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)ctx->n_cu * 8));
  // polar bins without arctan2 / the search over all edges where the tables apply (ODW_BIN_PLAIN=1: the plain kernel; tests
  // hold the two against each other and against numpy)
  const PhbBinAccel* accel = nullptr;
  PhbBinAccel A;
  if (polar && !getenv("ODW_BIN_PLAIN")) {
    ph_make_accel(A, polar, edges_a, n_a, edges_b, n_b);
    if (A.az_on || A.r_on) {
      if ((rc = ensure(ctx, ctx->ph_accel, sizeof A))) return rc;
      HIPCHK(ctx, hipMemcpyAsync(ctx->ph_accel.p, &A, sizeof A, hipMemcpyHostToDevice, ctx->stream));
      accel = (const PhbBinAccel*)ctx->ph_accel.p;
    }
  }
  const size_t lds = ph_bin_lds_bytes(n_a, n_b, nbins <= (uint64_t)kPhLdsBins ? nbins : 0, polar != 0);
  if (nbins <= (uint64_t)kPhLdsBins)
    hipLaunchKernelGGL((ph_bin_kernel<true>), dim3(grid), dim3(256), lds, ctx->stream, (const double*)ctx->ph_x.p,
                       (const double*)ctx->ph_y.p, m, origin[0], origin[1], (int)polar, (const double*)ctx->ph_edges.p,
                       (int)n_a, (const double*)ctx->ph_edges_b.p, (int)n_b, accel, (unsigned long long*)ctx->ph_counts.p);
  else
    hipLaunchKernelGGL((ph_bin_kernel<false>), dim3(grid), dim3(256), lds, ctx->stream, (const double*)ctx->ph_x.p,
                       (const double*)ctx->ph_y.p, m, origin[0], origin[1], (int)polar, (const double*)ctx->ph_edges.p,
                       (int)n_a, (const double*)ctx->ph_edges_b.p, (int)n_b, accel, (unsigned long long*)ctx->ph_counts.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(counts, ctx->ph_counts.p, nbins * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

// max and min of point . normal per candidate normal (odw_plane_screen)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2,fma", "default")))
#endif
static void plane_screen_blocks(const double* __restrict__ cloud, uint64_t n, const double* __restrict__ nx, const double* __restrict__ ny,
                                const double* __restrict__ nz, size_t nc, double* __restrict__ lo, double* __restrict__ hi) {
  constexpr size_t B = 8;
  size_t c0 = 0;
  for (; c0 + B <= nc; c0 += B) {
    double l[B], h[B], ax[B], ay[B], az[B];
    for (size_t k = 0; k < B; ++k) { l[k] = INFINITY; h[k] = -INFINITY; ax[k] = nx[c0 + k]; ay[k] = ny[c0 + k]; az[k] = nz[c0 + k]; }
    for (uint64_t p = 0; p < n; ++p) {
      const double x = cloud[3 * p], y = cloud[3 * p + 1], z = cloud[3 * p + 2];
      for (size_t k = 0; k < B; ++k) {
        const double a = x * ax[k] + y * ay[k] + z * az[k];
        l[k] = a < l[k] ? a : l[k];
        h[k] = a > h[k] ? a : h[k];
      }
    }
    for (size_t k = 0; k < B; ++k) { lo[c0 + k] = l[k]; hi[c0 + k] = h[k]; }
  }
  for (; c0 < nc; ++c0) {
    double l = INFINITY, h = -INFINITY;
    for (uint64_t p = 0; p < n; ++p) {
      const double a = cloud[3 * p] * nx[c0] + cloud[3 * p + 1] * ny[c0] + cloud[3 * p + 2] * nz[c0];
      l = a < l ? a : l;
      h = a > h ? a : h;
    }
    lo[c0] = l; hi[c0] = h;
  }
}

// ---- the screen of detectPlaneNormal's plane search, on the host (no device work) ----------------------------------
// extent[i * n_phi + j] = max - min over the cloud of (point . normal(phis[j], thetas[i])): the candidates of one grid
// of the search in the reference's order (meshgrid(phis, thetas) row-major).  Only a screen: the caller looks again,
// with numpy's own sums, at the candidates within rounding of the smallest extent.
int odw_plane_screen(const double* cloud, uint64_t n, const double* phis, int32_t n_phi, const double* thetas, int32_t n_theta,
                     double* extent) {
  if (!cloud || !phis || !thetas || !extent || n == 0 || n_phi < 1 || n_theta < 1 || n_phi > 4096 || n_theta > 4096)
    return ODW_ERR_INVALID;
  const size_t nc = (size_t)n_phi * (size_t)n_theta;
  for (uint64_t k = 0; k < 3 * n; ++k)
    if (!std::isfinite(cloud[k])) {                        // (the caller's margin test then keeps no candidate)
      for (size_t c = 0; c < nc; ++c) extent[c] = NAN;
      return ODW_OK;
    }
  // candidates in blocks of eight with their running minima and maxima in registers, the points inside: three loads
  // per point and block, no stores (candidates outside and arrays of minima in memory: 5 loads + 2 stores per product)
  std::vector<double> nx(nc), ny(nc), nz(nc), lo(nc), hi(nc);
  for (int i = 0; i < n_theta; ++i) {
    const double ct = std::cos(thetas[i]), st = std::sin(thetas[i]);
    for (int j = 0; j < n_phi; ++j) {
      const size_t c = (size_t)i * n_phi + j;
      nx[c] = std::cos(phis[j]) * st; ny[c] = std::sin(phis[j]) * st; nz[c] = ct;
    }
  }
  double* __restrict__ plo = lo.data();
  double* __restrict__ phi_ = hi.data();
  plane_screen_blocks(cloud, n, nx.data(), ny.data(), nz.data(), nc, plo, phi_);
  for (size_t c = 0; c < nc; ++c) extent[c] = phi_[c] - plo[c];
  return ODW_OK;
}

// the same screen for S clouds at once, each with its own grid (all grids of one size), on a small pool of threads of
// the library: the plane searches of a batch of scenes advance level by level, one call per level
namespace {
struct ScreenPool {
  std::mutex mu;
  std::condition_variable go, done;
  std::vector<std::thread> threads;
  std::function<void(int)> job;
  int next = 0, total = 0, finished = 0;
  uint64_t round = 0;
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lock(mu);
      go.wait(lock, [&] { return round != seen && next < total; });
      while (next < total) {
        const int k = next++;
        lock.unlock();
        job(k);
        lock.lock();
        if (++finished == total) done.notify_all();
      }
      seen = round;
    }
  }
  std::mutex callers;                      // (one job at a time: measuring threads of a sweep call in concurrently)
  void run(int n, std::function<void(int)> f) {
    std::lock_guard<std::mutex> one(callers);
    std::unique_lock<std::mutex> lock(mu);
    if (threads.empty()) {
      const unsigned hw = std::thread::hardware_concurrency();
      const int want = (int)std::max(1u, std::min(8u, hw ? hw / 2 : 2u));
      for (int k = 0; k < want; ++k) { threads.emplace_back([this] { worker(); }); threads.back().detach(); }
    }
    job = std::move(f);
    next = 0; total = n; finished = 0; ++round;
    go.notify_all();
    done.wait(lock, [&] { return finished == total; });
  }
};
ScreenPool* screen_pool() {
  static ScreenPool* p = new ScreenPool();      // (never destroyed: its threads may outlive main())
  return p;
}
}  // namespace

int odw_plane_screen_batch(const double* const* clouds, const uint64_t* n, int32_t n_clouds, const double* phis, int32_t n_phi,
                           const double* thetas, int32_t n_theta, double* extent) {
  if (!clouds || !n || !phis || !thetas || !extent || n_clouds < 1 || n_phi < 1 || n_theta < 1) return ODW_ERR_INVALID;
  std::vector<int> rc((size_t)n_clouds, 0);
  const size_t nc = (size_t)n_phi * (size_t)n_theta;
  uint64_t work = 0;
  for (int k = 0; k < n_clouds; ++k) work += n[k] * (uint64_t)nc;
  auto one = [&](int k) {
    rc[k] = (clouds[k] && n[k]) ? odw_plane_screen(clouds[k], n[k], phis + (size_t)k * n_phi, n_phi, thetas + (size_t)k * n_theta,
                                                   n_theta, extent + (size_t)k * nc)
                                : ODW_OK;
  };
  // (waking the pool costs ~0.1 ms: worth it for the 30 x 30 grids of a search's first level, not for its 10 x 10 ones)
  if (work < 400000 || n_clouds < 2) for (int k = 0; k < n_clouds; ++k) one(k);
  else screen_pool()->run(n_clouds, one);
  for (int k = 0; k < n_clouds; ++k)
    if (rc[k]) return rc[k];
  return ODW_OK;
}

int odw_hits_moments(odw_ctx* ctx, double* mean, double* var) {
  if (!ctx || !mean || !var) return fail(ctx, ODW_ERR_INVALID, "odw_hits_moments: bad argument");
  int rc = ph_need_selection(ctx, "odw_hits_moments");
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t m = ctx->ph_n;
  if (m == 0) return fail(ctx, ODW_ERR_INVALID, "odw_hits_moments: no rows selected");
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)ctx->n_cu * 8));
  std::vector<double> part;
  if (ctx->ph_moment_grid == grid && ctx->ph_moment_sums.size() == (size_t)grid * 6 + 3) {
    part = ctx->ph_moment_sums;          // a projection of this selection's points has added them up already
  } else {
    if ((rc = ensure(ctx, ctx->ph_part, ((size_t)grid * 6 + 3) * sizeof(double)))) return rc;
    part.resize((size_t)grid * 6 + 3);
    hipLaunchKernelGGL(ph_moment_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p,
                       (const uint32_t*)ctx->sort_vals[1].p, m, (double*)ctx->ph_part.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(part.data(), ctx->ph_part.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ph_moment_sums = part;
    ctx->ph_moment_grid = grid;
  }
  double s6[6] = {0, 0, 0, 0, 0, 0};
  for (unsigned b = 0; b < grid; ++b)
    for (int k = 0; k < 6; ++k) s6[k] += part[6 * (size_t)b + k];
  for (int k = 0; k < 3; ++k) {
    const double d = s6[k] / (double)m;                  // mean - provisional centre
    mean[k] = part[6 * (size_t)grid + k] + d;
    var[k] = std::max(0.0, s6[3 + k] / (double)m - d * d);
  }
  return ODW_OK;
}

}  // extern "C"

