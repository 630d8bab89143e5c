// odw_capi.hip -- C-ABI (include/odw_trace.h) over the gfx950 kernels.
//
// Host side of the native library: device context, scene/source upload into
// the HBM layouts of odw_device.h, BVH build for big scenes, inverse-CDF
// guide tables, launches on a private HIP stream, HIP-event timing, result
// fetch.  No torch types anywhere: plain pointers and sizes.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "odw_kernels.hip"
#include "odw_grid.hip"
#include "odw_mesh.hip"

using namespace odw;

namespace {

constexpr int kGuide = 1 << 16;
// Hit-list slots a wave reserves per atomic, and the smallest list that gets the room for it (A/B: ODW_HIT_BLOCK,
// ODW_HIT_BLOCK_MIN_ROWS).  Atomics on one address complete at about one per 3.6 ns on this chip whatever the number
// of waves, so a launch's length is bounded below by its atomic count: GettingStarted with a reservation per wave and
// recording step takes 0.23 / 0.66 ms for 1e6 / 3e6 rays, with blocks of 512 slots 0.16 / 0.34 (128: 0.19 / 0.45,
// 256: 0.17 / 0.36, 1024: 0.17 / 0.35, 4096: 0.33 / 0.68 -- the unused slots a wave tags at its end); at 1e8 rays
// 128: 10.7 ms, 256: 6.41, 512: 6.10, 1024: 6.06, 2048: 6.04, 4096: 6.13 (profiles/r03/r03q_hit_blocks.log).
static const uint32_t kHitBlock = [] { const char* e = getenv("ODW_HIT_BLOCK"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 64 ? (v & ~63) : 512); }();
static const uint64_t kHitBlockMinRows = [] { const char* e = getenv("ODW_HIT_BLOCK_MIN_ROWS"); return e ? (uint64_t)atoll(e) : (1ull << 16); }();
// room a list of `capacity` rows needs for reservations of `block` slots by `waves` waves: < 64 unused slots per
// block change, and the last block of every wave
static inline uint64_t hit_block_room(uint64_t capacity, uint64_t waves, uint64_t block) { return capacity * 64 / block + 64 + waves * block; }
constexpr int kPhiGuide = 1 << 8;        // azimuth table of a source (~1e2 knots)
constexpr int kSurfaceGuide = 1 << 10;  // per row of a surface sampler (tables of ~1e3 knots)
// Analytic scenes of up to this many primitives take the flat kernels (brute force over the primitives, scalar
// loads, one box test each).  Measured against the grid kernel's generic variant, which such scenes took above 16
// primitives until round 2: lens trains of 19 / 25 / 37 / 61 primitives 1.63e9 / 1.10e9 / 6.1e8 / 2.75e8 rays/s flat
// against 6.4e8 / 5.1e8 / 3.5e8 / 1.95e8 on the grid; random crowded scenes of 19 - 30 primitives 2 - 3.6 x faster
// (scripts/bench_lens_train.py, scripts/bench_crowded.py).  ODW_BVH_THRESHOLD (read when a context is created)
// overrides it: the tests keep the grid kernel's generic variant covered with 16.
constexpr int kBvhThreshold = 64;
const int kBvhLeaf = [] { const char* e = getenv("ODW_BVH_LEAF"); const int v = e ? atoi(e) : 0; return v > 0 && v < 200 ? v : 8; }();   // largest leaf the SAH may form (measured: 8 >= 4 > 2 > 1 on meshes; round 5, mesh kernel at 1e6 facets: 8 / 6 / 4 / 3 / 2 / 1 = 7.70 / 7.73 / 7.84 / 7.95 / 8.16 / 9.04 ms -- candidates per segment 18 -> 8, node visits 11.7 -> 14.7)
constexpr int kBvhSweepMax = 2048;       // nodes with more primitives use binned SAH

std::string g_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool host = false;       // allocated with malloc (a context without a device: odw_build_check)
};

}  // namespace

struct odw_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* up_pin = nullptr;                  // page-locked arena small uploads are staged in (upload())
  size_t up_off = 0;
  bool up_unstaged = false;
  bool host_only = false;                  // no device behind this context: buffers are host memory (odw_build_check)                // an upload since the last wait was copied from the caller's memory
  int n_cu = 256;
  std::string err;

  // host copies needed for lazy (re)builds
  std::vector<double> h_prim_f64;
  std::vector<int32_t> h_prim_i32;
  std::vector<int32_t> h_cond;            // prim | inside << 31
  std::vector<double> h_prim_hdr;         // 64-byte headers (boxes + the four integers), built with the BVH
  std::vector<char> h_dead;               // primitives no ray can meet (no face, or an empty box)
  std::vector<double> h_group_f64, h_group_gdir;
  std::vector<int32_t> h_group_i32;
  std::vector<uint64_t> h_seq;
  // scene-compiled flat kernel (odw_spec.hip)
  int compile_mode = 0;                    // ODW_COMPILE_*: sticky, applies to every scene uploaded later too
  bool spec_dirty = true;                  // scene / limits changed since the last binding attempt
  hipFunction_t spec_fn = nullptr;         // bound kernel (owned by the process-wide cache), or null
  bool spec_lean = false, spec_stoch = false;
  // ODW_COMPILE_AUTO: the scene's kernel is not there yet (not hot enough, or being compiled)
  bool spec_pending = false;
  std::string spec_key;
  uint64_t spec_hot_rays = 50000000;       // rays traced with a structure on generic kernels (process-wide count) after
                                           // which its compilation starts (ODW_SPEC_HOT_RAYS at odw_create)
  double spec_seconds = 0;                 // compile time of the bound kernel (0: it came from a cache)
  int spec_cache_hit = 0;                  // 0 compiled now, 1 process cache, 2 disk cache
  bool have_scene = false, have_source = false, have_limits = false;
  bool bvh_dirty = true;
  int flat_limit = kBvhThreshold;          // most primitives the flat kernels take (ODW_BVH_THRESHOLD at odw_create)
  bool lean = false;                       // no grating group, no finite absorption length: LEAN kernels

  DevBuf prim_f64, prim_hdr, prim_i32, cond_i32, group_f64, group_i32, group_gdir, seq_mask;
  DevBuf bvh_nodes, bvh_prims, tri_nrm;
  DevBuf bvh_leaf, bvh_wide;                         // leaf records and eight-wide tree of the mesh kernel (odw_mesh.hip)
  DevBuf grid_bounds, grid_cells, grid_items, dbg;   // rectilinear grid of big analytic scenes (odw_grid.hip)
  DevBuf phi_tab, t_tab, t_guide, phi_guide, d_source, d_det;
  DeviceSource h_source;
  DeviceDetector h_det;
  DevBuf hits, hit_count, chunk_counter;
  // ONE block holds what ranks sum: [kResultsHead words: the ODW_CNT_COUNT counters, padded] [n_bins histogram words]
  // -- a multi-GPU job reduces it with a single collective (odw_device_results); `counters` and `hist` are views into it
  DevBuf results, hist, counters;
  DevBuf segs, seg_count;                  // RecordRays segment list
  uint64_t seg_capacity = 0;
  DevBuf ray_o, ray_d, ray_p, ray_aos, samp_t, samp_phi;
  DevBuf sort_keys[2], sort_vals[2], sort_tmp, sorted_rows;
  // post-hoc binning of the rows in HBM (odw_posthoc.hip): the selection = sort_vals[1][0 .. ph_n)
  DevBuf ph_sel_entering, ph_flags, ph_x, ph_y, ph_sorted, ph_small, ph_part, ph_edges, ph_edges_b, ph_counts, ph_sel_hist;
  DevBuf ph_accel;                                   // tables of the polar binning (odw_posthoc.hip: PhbBinAccel)
  DevBuf ph_bitmap, ph_before, ph_row_of;            // ordered selection without a sort (odw_posthoc.hip: ph_mark_kernel)
  uint64_t alt_hit_ray_end = 0;    // the same for the list odw_swap_hit_lists has put aside
  uint64_t hit_ray_end = 0;        // ray indices of the rows in the hit list lie below this (0: list empty; 1 << 48: unknown)
  uint64_t hit_ray_begin = 0, alt_hit_ray_begin = 0;   // ... and at or above this (meaningful while hit_ray_end is a real bound)
  uint64_t ph_n = 0, ph_n_entering = 0;
  int ph_group = -1;
  bool ph_valid = false, ph_projected = false, ph_entering_built = false;
  std::vector<double> ph_moment_sums;      // per-block moment sums + centre of the current selection's points (odw_hits_project, odw_hits_moments)
  unsigned ph_moment_grid = 0;             // 0: none
  // stochastic surfaces: one table set per sampler, descriptor block, (group, kind) -> index
  struct SurfaceBufs { DevBuf phi_tab, t_tab, t_guide, atom_mass; };
  std::vector<SurfaceBufs> surf_bufs;
  DevBuf d_samplers, d_group_sampler;
  int n_samplers = 0;
  uint64_t surface_seed = 0;
  // surface source (emitter) tables + the explicit-ray staging of its launches
  DevBuf em_prim_f64, em_prim_i32, em_cond, em_face_i32, em_face_cdf, em_t_tab, em_t_guide, em_o, em_d, em_tri_nrm;
  DeviceEmitter h_emitter;
  bool emitter_active = false;   // the most recently uploaded source is a surface source
  uint64_t hit_capacity = 0, n_bins = 0;   // hit_capacity: rows the caller asked for
  uint64_t hit_slots = 0;                  // rows allocated (capacity + slack for block reservations)
  // batch launches (odw_upload_scene_batch / odw_trace_batch): the value tables of batch_n scenes of one structure side by
  // side, one segment of the batch's hit list and one pair of counters per scene.  odw_batch_select makes a segment the
  // context's hit list (ctx->hits / hit_count become views; the context's own list waits in own_*)
  DevBuf batch_values, batch_hits, batch_hit_count, batch_rays_buf;
  int batch_n = 0;                         // scenes of the uploaded batch (0: none)
  size_t batch_prims = 0;                  // primitives per scene of that batch
  bool batch_launch = false;               // launch_trace: this launch is a batch
  uint64_t batch_stride = 0;               // doubles per scene block
  uint64_t batch_seg_slots = 0, batch_seg_capacity = 0, batch_rays = 0, batch_first = 0;
  int batch_traced = 0;                    // scenes of the last odw_trace_batch
  bool batch_rows_ok = false;              // its segments hold rows: the launch recorded hits and was issued without error
  bool batch_marked = false;               // ... whose rows noted their slots in phb_row_of while they were recorded
  bool batch_pts = false;                  // ... and their points in phb_pts (the chain's projection reads those)
  int batch_selected = -1;
  std::string batch_spec_text;             // the structure all scenes of the batch share (compiled kernels)
  hipFunction_t spec_batch_fn = nullptr;   // the scene-compiled kernel's BATCH variant (bound on the first batch launch)
  bool spec_batch_failed = false;          // ... could not be built for the bound structure: generic kernels for its batches
  DevBuf own_hits, own_hit_count;
  uint64_t own_capacity = 0, own_slots = 0, own_ray_begin = 0, own_ray_end = 0;
  // a run's rows kept in HBM beyond the launches that recorded them (odw_archive_append / odw_archive_select)
  DevBuf archive, archive_count;
  uint64_t archive_slots = 0, archive_unused = 0, archive_ray_begin = 0, archive_ray_end = 0;
  bool archive_selected = false;
  // post-hoc binning of all segments at once (odw_batch_hits_*, odw_posthoc.hip): per-scene slices of these
  DevBuf phb_row_of, phb_words, phb_sel, phb_small, phb_rows, phb_x, phb_y, phb_part, phb_sel_hist, phb_cand, phb_counts;
  DevBuf phb_scenes, phb_hist, phb_planes, phb_strides, phb_origins, phb_accel, phb_pts;   // device-resident state of the chain (odw_posthoc_batch.hip)
  void* phb_pin_p = nullptr;               // page-locked block the chain's results arrive in
  size_t phb_pin_bytes = 0;
  hipEvent_t phb_ev = nullptr;             // end of the piece enqueued last (odw_batch_hits_begin / _measure)
  int phb_stage = 0, phb_S = 0;            // 1 begin enqueued, 2 sampled, 3 measure enqueued, 4 measured
  uint64_t phb_cap = 0, phb_nbins = 0, phb_keep = 0;
  size_t phb_part_stride = 0;
  std::vector<double> phb_edges_host;      // the edges on the device (uploaded when they change)
  int phb_edges_na = 0, phb_edges_nb = 0, phb_edges_polar = -1;
  std::vector<uint64_t> phb_used, phb_n, phb_leaving;
  std::vector<int32_t> phb_ordered;
  std::vector<char> phb_on;
  uint64_t phb_xy_stride = 0;
  int phb_group = -1;
  bool phb_valid = false, phb_projected = false;
  // second hit list (odw_swap_hit_lists): while one is traced into, the other is copied to the host
  // on a stream of its own
  DevBuf alt_hits, alt_hit_count;
  uint64_t alt_capacity = 0, alt_slots = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t alt_ready = nullptr;          // recorded on the trace stream when the list was put aside
  bool swapping = false;                   // lists are swapped: appends stay dense (no block reservations)

  TraceParams P;
  odw_detector_desc det_desc;

  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
  double timing_ms = 0;
  uint64_t timing_launches = 0;
};

namespace {

int fail(odw_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  g_error = msg;
  return code;
}

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(ctx, ODW_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

int ensure(odw_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.bytes >= bytes && b.p) return ODW_OK;
  if (ctx && ctx->host_only) {
    // a context without a device (odw_build_check): the tables the builders "upload" live in host memory -- the same code
    // paths, under a CPU sanitizer
    if (b.p) free(b.p);
    b.p = malloc(bytes);
    b.bytes = b.p ? bytes : 0;
    b.host = true;
    return b.p ? ODW_OK : fail(ctx, ODW_ERR_DEVICE, "out of host memory");
  }
  if (b.p) HIPCHK(ctx, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  HIPCHK(ctx, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return ODW_OK;
}

// Host to device on the context's stream.  Small tables (a scene's, a batch's: a few KB) travel through a page-locked
// arena of the context and are NOT waited for: a copy from pageable memory blocks the caller until it has run, and in a
// sweep it runs behind whatever other contexts have queued on the copy path -- up to a launch's length (10 ms stalls in
// the launch of a group, measured).  The caller's array may go away at once either way; upload_done() is the wait for
// copies that did not fit the arena.
constexpr size_t kUploadArena = 4u << 20, kUploadStaged = 512u << 10;
int upload(odw_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  int rc = ensure(ctx, b, bytes);
  if (rc) return rc;
  if (!bytes) return ODW_OK;
  if (ctx->host_only) { std::memcpy(b.p, src, bytes); return ODW_OK; }
  if (bytes <= kUploadStaged) {
    if (!ctx->up_pin) {
      if (hipHostMalloc(&ctx->up_pin, kUploadArena, hipHostMallocDefault) != hipSuccess) { ctx->up_pin = nullptr; (void)hipGetLastError(); }
      ctx->up_off = 0;
    }
    if (ctx->up_pin) {
      const size_t need = (bytes + 63) & ~(size_t)63;
      if (ctx->up_off + need > kUploadArena) {             // the arena comes round: what lies in it must have been copied
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->up_off = 0;
      }
      char* at = (char*)ctx->up_pin + ctx->up_off;
      std::memcpy(at, src, bytes);
      ctx->up_off += need;
      HIPCHK(ctx, hipMemcpyAsync(b.p, at, bytes, hipMemcpyHostToDevice, ctx->stream));
      return ODW_OK;
    }
  }
  HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ctx->up_unstaged = true;
  return ODW_OK;
}
// after a series of uploads from arrays that are about to go away: waits only if one of them was copied in place
int upload_done(odw_ctx* ctx) {
  if (ctx->up_unstaged) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->up_unstaged = false;
  }
  return ODW_OK;
}

void release(DevBuf& b) {
  if (b.p && b.host) free(b.p);
  else if (b.p) (void)hipFree(b.p);
  b.host = false;
  b.p = nullptr;
  b.bytes = 0;
}

constexpr size_t kResultsHead = 16;      // words in front of the histogram (128 B: the bins keep their alignment)
static_assert(ODW_CNT_COUNT <= kResultsHead, "counters must fit the head of the results block");

// the results block for a histogram of n_bins bins; the counters survive a reallocation
int ensure_results(odw_ctx* ctx, uint64_t n_bins);
// ctx->hits / hit_count are views of a batch segment (odw_batch_select): give the context its own list back
void batch_unselect(odw_ctx* ctx);
// room for the post-hoc chain of a batch (odw_posthoc_batch.hip)
int phb_reserve(odw_ctx* ctx, int S, uint64_t rays_per_scene, uint64_t slots);

// ---- primitive bounding boxes in global coordinates -----------------------
void local_bounds(int type, const double* par, double lo[3], double hi[3]) {
  switch (type) {
    case ODW_PRIM_BOX:
      lo[0] = lo[1] = lo[2] = 0; hi[0] = par[0]; hi[1] = par[1]; hi[2] = par[2];
      break;
    case ODW_PRIM_SPHERE:
      for (int i = 0; i < 3; ++i) { lo[i] = -par[0]; hi[i] = par[0]; }
      break;
    case ODW_PRIM_CYLINDER:
      lo[0] = lo[1] = -par[0]; hi[0] = hi[1] = par[0]; lo[2] = 0; hi[2] = par[1];
      break;
    case ODW_PRIM_CONE: {
      const double r = std::max(par[0], par[1]);
      lo[0] = lo[1] = -r; hi[0] = hi[1] = r; lo[2] = 0; hi[2] = par[2];
      break;
    }
    case ODW_PRIM_PARABOLOID: {
      const double r = 2.0 * std::sqrt(std::max(par[0] * par[1], 0.0));
      lo[0] = lo[1] = -r; hi[0] = hi[1] = r; lo[2] = 0; hi[2] = par[1];
      break;
    }
    default: {
      const double r = par[0] + par[1];
      lo[0] = lo[1] = -r; hi[0] = hi[1] = r; lo[2] = -par[1]; hi[2] = par[1];
    }
  }
}

struct Box {
  double lo[3], hi[3];
  void reset() { for (int i = 0; i < 3; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; } }
  void grow(const Box& o) {
    for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], o.lo[i]); hi[i] = std::max(hi[i], o.hi[i]); }
  }
};

Box world_box(const double* pf, int type, double slack) {
  Box b;
  b.reset();
  if (type == ODW_PRIM_TRIANGLE) {   // v0, e1, e2 in global coordinates
    for (int i = 0; i < 3; ++i) {
      const double a = pf[i], c1 = pf[i] + pf[3 + i], c2 = pf[i] + pf[6 + i];
      const double s = slack + 1e-9 * (std::fabs(a) + std::fabs(c1) + std::fabs(c2));
      b.lo[i] = std::min(a, std::min(c1, c2)) - s;
      b.hi[i] = std::max(a, std::max(c1, c2)) + s;
    }
    return b;
  }
  double lo[3], hi[3];
  local_bounds(type, pf + 12, lo, hi);
  for (int c = 0; c < 8; ++c) {
    const double l[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
    // global = R^T (local - t)
    const double d[3] = {l[0] - pf[3], l[1] - pf[7], l[2] - pf[11]};
    const double g[3] = {pf[0] * d[0] + pf[4] * d[1] + pf[8] * d[2],
                         pf[1] * d[0] + pf[5] * d[1] + pf[9] * d[2],
                         pf[2] * d[0] + pf[6] * d[1] + pf[10] * d[2]};
    for (int i = 0; i < 3; ++i) { b.lo[i] = std::min(b.lo[i], g[i]); b.hi[i] = std::max(b.hi[i], g[i]); }
  }
  for (int i = 0; i < 3; ++i) {
    const double s = slack + 1e-9 * (std::fabs(b.lo[i]) + std::fabs(b.hi[i]));
    b.lo[i] -= s;
    b.hi[i] += s;
  }
  return b;
}

// BVH node, 64 bytes = one cache line: the boxes of BOTH children in float32
// (rounded outward), so one fetch decides where to go next.  child >= 0: inner
// node index; count > 0: leaf = `count` primitives from bvh_prims[child].
struct BvhNode {
  float lo0[3], hi0[3], lo1[3], hi1[3];
  int32_t child0, child1, count0, count1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be one 64-byte line");

float round_down(double v) {
  float f = (float)v;
  return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
float round_up(double v) {
  float f = (float)v;
  return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}

// Surface-area-heuristic build (full sweep on the three axes).  Measured on
// hugeArray: 33 node visits and 3.0 primitive tests per segment against 57 /
// 5.8 with median splits.
struct BvhBuilder {
  const std::vector<Box>& boxes;
  std::vector<int> order;       // leaf primitive order
  std::vector<BvhNode> nodes;
  int max_depth = 0;
  // from this depth on only median splits down to leaves of kBvhLeaf: whatever
  // the SAH did above, the tree stays within the traversal stack
  int balanced_depth;

  // The heuristic goes on below balanced_depth wherever the levels that are left still hold a median-split subtree of
  // the node's primitives (round 5; ODW_BVH_SAH_DEEP=0: medians from balanced_depth on, as before).  Ball lens of 1e6
  // facets under the mesh kernel: 12.8 -> 7.2 candidate facets per segment, 6.02 -> 5.77 ms per 1e7 rays, build 0.8 -> 1.1 s
  // (full sweeps only up to 256 primitives: above that, 32 bins).
  bool deep_sah = true;
  int sweep_max = kBvhSweepMax;

  explicit BvhBuilder(const std::vector<Box>& b) : boxes(b) {
    const double n = (double)std::max<size_t>(b.size(), 8);
    balanced_depth = std::max(2, ODW_BVH_STACK - 2 - (int)std::ceil(std::log2(n / kBvhLeaf)) - 1);
    const char* e = getenv("ODW_BVH_SAH_DEEP");
    deep_sah = !(e && e[0] == '0');
    if (deep_sah) sweep_max = 256;
  }
  bool sah_ok(int depth, int m) const {
    if (!deep_sah) return depth < balanced_depth;
    const int need = (int)std::ceil(std::log2(std::max(1.0, (double)m / kBvhLeaf)));
    return depth + need + 2 <= ODW_BVH_STACK - 3;
  }

  static double area(const Box& b) {
    const double ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    return 2.0 * (ex * ey + ey * ez + ez * ex);
  }

  struct Ref { int32_t child, count; Box box; };

  // builds the subtree over ids; returns either a leaf ref or an inner node ref
  Ref build(std::vector<int>& ids, int depth) {
    max_depth = std::max(max_depth, depth);
    Box bb;
    bb.reset();
    for (int i : ids) bb.grow(boxes[i]);
    const int m = (int)ids.size();
    auto make_leaf = [&]() {
      Ref r;
      r.child = (int32_t)order.size();
      r.count = m;
      r.box = bb;
      for (int i : ids) order.push_back(i);
      return r;
    };
    if (m <= 1) return make_leaf();
    const bool sah = sah_ok(depth, m);
    if (!sah && m <= kBvhLeaf) return make_leaf();
    if (m > sweep_max || !sah) return build_big(ids, depth, bb);
    // SAH sweep
    double best_cost = INFINITY;
    int best_axis = -1, best_split = 0;
    std::vector<int> sorted(ids), best_sorted;
    std::vector<double> right_area(m);
    for (int a = 0; a < 3; ++a) {
      std::sort(sorted.begin(), sorted.end(), [&](int x, int y) {
        const double cx = boxes[x].lo[a] + boxes[x].hi[a], cy = boxes[y].lo[a] + boxes[y].hi[a];
        return cx < cy || (cx == cy && x < y);
      });
      Box r;
      r.reset();
      for (int i = m - 1; i > 0; --i) { r.grow(boxes[sorted[i]]); right_area[i] = area(r); }
      Box l;
      l.reset();
      for (int i = 1; i < m; ++i) {
        l.grow(boxes[sorted[i - 1]]);
        const double cost = area(l) * i + right_area[i] * (m - i);
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = i; best_sorted = sorted; }
      }
    }
    // leaf if splitting does not pay (traversal step ~ 1 primitive test) and it is small
    const double leaf_cost = area(bb) * m;
    if (m <= kBvhLeaf && best_cost + area(bb) >= leaf_cost) return make_leaf();
    if (best_axis < 0) return make_leaf();
    std::vector<int> left(best_sorted.begin(), best_sorted.begin() + best_split);
    std::vector<int> right(best_sorted.begin() + best_split, best_sorted.end());
    return inner(left, right, depth, bb);
  }

  // big nodes (meshes): binned SAH over 32 bins of the centroid range, O(m) per
  // node; from balanced_depth on: median splits, which bound the remaining
  // depth by log2(m / kBvhLeaf)
  Ref build_big(std::vector<int>& ids, int depth, const Box& bb) {
    const int m = (int)ids.size();
    Box cb;
    cb.reset();
    for (int i : ids)
      for (int a = 0; a < 3; ++a) {
        const double c = boxes[i].lo[a] + boxes[i].hi[a];
        cb.lo[a] = std::min(cb.lo[a], c);
        cb.hi[a] = std::max(cb.hi[a], c);
      }
    int axis = 0;
    for (int a = 1; a < 3; ++a) if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
    auto centroid = [&](int i, int a) { return boxes[i].lo[a] + boxes[i].hi[a]; };
    std::vector<int> left, right;
    bool split_done = false;
    if (sah_ok(depth, m) && cb.hi[axis] > cb.lo[axis]) {
      constexpr int kBins = 32;
      double best_cost = INFINITY;
      int best_axis = -1, best_bin = 0;
      for (int a = 0; a < 3; ++a) {
        const double ext = cb.hi[a] - cb.lo[a];
        if (!(ext > 0)) continue;
        Box bins[kBins];
        int cnt[kBins] = {0};
        for (auto& b : bins) b.reset();
        for (int i : ids) {
          const int k = std::min(kBins - 1, (int)((centroid(i, a) - cb.lo[a]) / ext * kBins));
          bins[k].grow(boxes[i]);
          ++cnt[k];
        }
        double ra[kBins];
        int rc[kBins];
        Box r;
        r.reset();
        int c = 0;
        for (int k = kBins - 1; k > 0; --k) { r.grow(bins[k]); c += cnt[k]; ra[k] = c ? area(r) : 0.0; rc[k] = c; }
        Box l;
        l.reset();
        c = 0;
        for (int k = 1; k < kBins; ++k) {
          l.grow(bins[k - 1]);
          c += cnt[k - 1];
          if (c == 0 || rc[k] == 0) continue;
          const double cost = area(l) * c + ra[k] * rc[k];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
        }
      }
      if (best_axis >= 0) {
        const double ext = cb.hi[best_axis] - cb.lo[best_axis];
        for (int i : ids) {
          const int k = std::min(kBins - 1, (int)((centroid(i, best_axis) - cb.lo[best_axis]) / ext * kBins));
          (k < best_bin ? left : right).push_back(i);
        }
        split_done = !left.empty() && !right.empty();
      }
    }
    if (!split_done) {   // median split along the widest centroid axis
      std::vector<int> sorted(ids);
      std::nth_element(sorted.begin(), sorted.begin() + m / 2, sorted.end(), [&](int x, int y) {
        const double cx = centroid(x, axis), cy = centroid(y, axis);
        return cx < cy || (cx == cy && x < y);
      });
      left.assign(sorted.begin(), sorted.begin() + m / 2);
      right.assign(sorted.begin() + m / 2, sorted.end());
    }
    return inner(left, right, depth, bb);
  }

  Ref inner(std::vector<int>& left, std::vector<int>& right, int depth, const Box& bb) {
    const int id = (int)nodes.size();
    nodes.emplace_back();
    const Ref l = build(left, depth + 1);
    const Ref r = build(right, depth + 1);
    BvhNode& nd = nodes[id];
    for (int k = 0; k < 3; ++k) {
      nd.lo0[k] = round_down(l.box.lo[k]); nd.hi0[k] = round_up(l.box.hi[k]);
      nd.lo1[k] = round_down(r.box.lo[k]); nd.hi1[k] = round_up(r.box.hi[k]);
    }
    nd.child0 = l.child; nd.count0 = l.count;
    nd.child1 = r.child; nd.count1 = r.count;
    Ref out;
    out.child = id;
    out.count = 0;
    out.box = bb;
    return out;
  }
};


// ---- eight-wide tree of the mesh kernel (odw_mesh.hip) --------------------------------------
// The binary tree above, collapsed: a wide node takes up to eight descendants of a binary node (the one with the
// largest box is opened next; one whose subtree is too high for the levels that remain goes first -- that bounds
// the depth, and with one stack entry per level the traversal stack, at kWideMaxDepth + 1).  The children's boxes
// are stored as 8-bit offsets from the node's corner in units of a power of two per axis (rounded outward);
// children sit in the slot whose sign pattern (x, y, z: away from / towards the corner) fits the direction from
// the node's centre to theirs best, so that `slot XOR ray octant` orders them roughly front to back without a
// sort.  Inner children are consecutive nodes (slot order), the facets of leaf children consecutive leaf
// records (slot order, <= 15 per leaf).
// Node = 32 words (128 bytes, 20 used):
//   0..2 corner (float)            3  exponent bytes x | y << 8 | z << 16 (biased: scale = 2^(e - 127))
//   4    first inner child         5  first leaf record
//   6    inner slots | leaf slots << 8          7  facets per leaf slot (4 bits each)
//   8..13 near corner offsets: x of slots 0-3, x of 4-7, y, y, z, z     14..19 far corner offsets, the same way
//   20..23 the solid every primitive below a slot belongs to (16 bits per slot, 0xffff: several or none): a ray that
//          has just left a convex solid drops the slots of that solid before it looks at their boxes' order
//   24..31 per slot, the cone of the outward normals of the facets below it, where they all belong to ONE STRICTLY
//          CONVEX solid: bytes 0..2 an axis a = round(127 u) (signed), byte 3 a threshold T + 3 <= 126 (signed); no cone:
//          0, 0, 0, 127.  A ray that travels INSIDE that solid (it entered through one of its facets, odw_mesh.hip `inside`)
//          can only leave through facets it meets from behind, d . n > 0; the kernel drops a slot when
//          v_dot4(word, [round(127 d), 127]) < 0, i.e. round(127 d) . a < -127 (T + 3): then d . a < -(T + 1.5) whatever the
//          rounding of d did (|round(127 d) - 127 d| <= 0.5 per axis, |a|_1 <= 220: 110 of the 190 to spare), and with
//          T = ceil(|a| sin(widest angle between a and a normal + asin(cone_margin))) every facet below the slot has
//          d . n < -cone_margin -- the whole neighbourhood of the facet the ray starts on, for one.  cone_margin
//          (WideBvh::margin) is what keeps the rule exact: the start point lies on its facet up to the closed-edge slack, so
//          it is above the plane of a dropped facet by less than `above`, and the plane would be met at
//          t < above / margin <= dist_tol, where consider() rejects it anyway.
constexpr int kWideWords = 32;
constexpr int kWideMaxDepth = 11;

struct WideBvh {
  struct Ref { int32_t child, count; float lo[3], hi[3]; };     // count > 0: leaf of `count` primitives from order[child]
  const std::vector<BvhNode>& bn;
  const std::vector<int>& order;
  const std::vector<int>& solid_of;           // solid id of every primitive
  const float* out_normal = nullptr;          // 3 per primitive: outward unit normal of the facets of convex solids, NaN for the rest
  double margin = 1.0;                        // >= 0.5: no cones
  std::vector<int> span_lo, span_hi;          // per binary node: its primitives are order[span_lo .. span_hi)
  std::vector<int> height;
  std::vector<int> solid_below;               // per binary node: the one solid of its primitives, -1 several, -2 not asked yet
  std::vector<uint32_t> nodes;
  std::vector<int> leaf_prim;                 // primitive of every leaf record
  std::vector<float> leaf_center;             // 3 per record: the centre its group is expressed around
  int depth = 0;
  bool ok = true;

  WideBvh(const std::vector<BvhNode>& n, const std::vector<int>& o, const std::vector<int>& so)
      : bn(n), order(o), solid_of(so), height(n.size(), -1), solid_below(n.size(), -2) {}

  int ref_solid(const Ref& r) {
    if (r.count == 0) return node_solid(r.child);
    int s = solid_of[order[(size_t)r.child]];
    for (int k = 1; k < r.count; ++k)
      if (solid_of[order[(size_t)r.child + k]] != s) return -1;
    return s;
  }
  int node_solid(int n) {
    if (solid_below[n] != -2) return solid_below[n];
    const BvhNode& nd = bn[n];
    int s = -3;                                 // nothing seen yet
    for (const Ref& r : {ref0(nd), ref1(nd)}) {
      if (far_box(r.lo)) continue;
      const int c = ref_solid(r);
      s = s == -3 ? c : (s == c ? s : -1);
    }
    return solid_below[n] = s == -3 ? -1 : s;
  }

  // (leaves are written to `order` in the order the builder meets them: a subtree's primitives are one run of it)
  void node_span(int n, int& lo, int& hi) {
    if (span_lo.empty()) { span_lo.assign(bn.size(), -1); span_hi.assign(bn.size(), -1); }
    if (span_lo[n] < 0) {
      int l = INT32_MAX, h = 0;
      const BvhNode& nd = bn[n];
      for (const Ref& r : {ref0(nd), ref1(nd)}) {
        if (far_box(r.lo)) continue;
        int a, b;
        if (r.count > 0) { a = r.child; b = r.child + r.count; } else node_span(r.child, a, b);
        l = std::min(l, a); h = std::max(h, b);
      }
      span_lo[n] = l == INT32_MAX ? 0 : l; span_hi[n] = h;
    }
    lo = span_lo[n]; hi = span_hi[n];
  }
  // the cone word of a slot (see the node layout above)
  uint32_t cone_word(const Ref& r) {
    constexpr uint32_t none = 0x7f000000u;
    if (!out_normal || !(margin < 0.5)) return none;
    int lo, hi;
    if (r.count > 0) { lo = r.child; hi = r.child + r.count; } else node_span(r.child, lo, hi);
    double sum[3] = {0.0, 0.0, 0.0};
    for (int k = lo; k < hi; ++k) {
      const float* nv = out_normal + 3 * (size_t)order[(size_t)k];
      if (!(nv[0] == nv[0])) return none;
      for (int a = 0; a < 3; ++a) sum[a] += (double)nv[a];
    }
    const double len = std::sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
    if (!(len > 1e-6 * (double)(hi - lo)) || hi <= lo) return none;
    int ax[3];
    double al = 0.0;
    for (int a = 0; a < 3; ++a) { ax[a] = (int)std::lround(127.0 * sum[a] / len); al += (double)ax[a] * ax[a]; }
    al = std::sqrt(al);
    if (!(al > 100.0)) return none;
    double cmin = 1.0;
    for (int k = lo; k < hi; ++k) {
      const float* nv = out_normal + 3 * (size_t)order[(size_t)k];
      const double nl = std::sqrt((double)nv[0] * nv[0] + (double)nv[1] * nv[1] + (double)nv[2] * nv[2]);
      cmin = std::min(cmin, ((double)nv[0] * ax[0] + (double)nv[1] * ax[1] + (double)nv[2] * ax[2]) / (al * nl));
    }
    // (1e-5: the normals are float32 copies of unit vectors, the ray's direction is rounded to float32 in the kernel)
    const double theta = std::acos(std::max(-1.0, std::min(1.0, cmin))) + std::asin(margin) + 1e-5;
    if (!(theta < 1.5)) return none;
    const double t = std::ceil(al * std::sin(theta));
    if (!(t + 3.0 <= 126.0)) return none;                      // (cones that wide drop next to nothing)
    return (uint32_t)(ax[0] & 0xff) | ((uint32_t)(ax[1] & 0xff) << 8) | ((uint32_t)(ax[2] & 0xff) << 16) | ((uint32_t)(t + 3.0) << 24);
  }

  static bool far_box(const float* lo) { return lo[0] >= 3.0e38f; }        // the child a wrapper root does not have
  static Ref ref0(const BvhNode& nd) { Ref r{nd.child0, nd.count0, {nd.lo0[0], nd.lo0[1], nd.lo0[2]}, {nd.hi0[0], nd.hi0[1], nd.hi0[2]}}; return r; }
  static Ref ref1(const BvhNode& nd) { Ref r{nd.child1, nd.count1, {nd.lo1[0], nd.lo1[1], nd.lo1[2]}, {nd.hi1[0], nd.hi1[1], nd.hi1[2]}}; return r; }

  int node_height(int n) {
    if (height[n] >= 0) return height[n];
    const BvhNode& nd = bn[n];
    int h = 0;
    if (nd.count0 == 0 && !far_box(nd.lo0)) h = std::max(h, node_height(nd.child0));
    if (nd.count1 == 0 && !far_box(nd.lo1)) h = std::max(h, node_height(nd.child1));
    return height[n] = h + 1;
  }
  static double area(const Ref& r) {
    const double ex = (double)r.hi[0] - r.lo[0], ey = (double)r.hi[1] - r.lo[1], ez = (double)r.hi[2] - r.lo[2];
    return 2.0 * (ex * ey + ey * ez + ez * ex);
  }

  void build() {
    if (bn.empty()) { ok = false; return; }
    if (node_height(0) > 3 * (kWideMaxDepth + 1)) { ok = false; return; }
    nodes.assign(kWideWords, 0u);
    fill(0, 0, 0);
  }

  void fill(size_t index, int n, int d) {
    depth = std::max(depth, d);
    if (d > kWideMaxDepth) { ok = false; return; }
    std::vector<Ref> cand;
    for (const Ref& r : {ref0(bn[n]), ref1(bn[n])})
      if (!far_box(r.lo)) cand.push_back(r);
    const int allowed = 3 * (kWideMaxDepth - d);             // binary height a child's subtree may have
    while (cand.size() < 8) {
      int pick = -1;
      int tallest = allowed;
      for (size_t k = 0; k < cand.size(); ++k)
        if (cand[k].count == 0 && node_height(cand[k].child) > tallest) { tallest = node_height(cand[k].child); pick = (int)k; }
      if (pick < 0) {
        double best = -1.0;
        for (size_t k = 0; k < cand.size(); ++k)
          if (cand[k].count == 0 && area(cand[k]) > best) { best = area(cand[k]); pick = (int)k; }
      }
      if (pick < 0) break;                                     // leaves only
      const BvhNode& nd = bn[cand[pick].child];
      cand[pick] = ref0(nd);
      cand.push_back(ref1(nd));
    }
    // the node's box and the slots
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const Ref& r : cand)
      for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], r.lo[a]); hi[a] = std::max(hi[a], r.hi[a]); }
    int slot_of[8], cand_in[8];
    for (int k = 0; k < 8; ++k) { slot_of[k] = -1; cand_in[k] = -1; }
    {
      struct Pair { double cost; int c, s; };
      std::vector<Pair> pairs;
      for (size_t c = 0; c < cand.size(); ++c)
        for (int sl = 0; sl < 8; ++sl) {
          double cost = 0.0;
          for (int a = 0; a < 3; ++a) {
            const double v = 0.5 * ((double)cand[c].lo[a] + cand[c].hi[a]) - 0.5 * ((double)lo[a] + hi[a]);
            cost += ((sl >> a) & 1) ? v : -v;
          }
          pairs.push_back({cost, (int)c, sl});
        }
      std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.cost > y.cost; });
      for (const Pair& pr : pairs)
        if (slot_of[pr.c] < 0 && cand_in[pr.s] < 0) { slot_of[pr.c] = pr.s; cand_in[pr.s] = pr.c; }
    }
    uint32_t w[kWideWords] = {0};
    uint32_t ebyte[3];
    double scale[3];
    for (int a = 0; a < 3; ++a) {
      std::memcpy(&w[a], &lo[a], 4);
      const double ext = (double)hi[a] - (double)lo[a];
      int e = -100;
      if (ext > 0) {
        int ex2;
        std::frexp(ext / 255.0, &ex2);                        // ext / 255 = m 2^ex2, 0.5 <= m < 1: 2^ex2 >= ext / 255
        e = ex2;
      }
      e = std::max(-126, std::min(127, e));
      while (std::ldexp(255.0, e) < ext && e < 127) ++e;
      ebyte[a] = (uint32_t)(e + 127);
      scale[a] = std::ldexp(1.0, e);
    }
    w[3] = ebyte[0] | (ebyte[1] << 8) | (ebyte[2] << 16);
    uint32_t imask = 0, lmask = 0, counts = 0;
    const uint32_t child_base = (uint32_t)(nodes.size() / kWideWords);
    const uint32_t leaf_base = (uint32_t)leaf_prim.size();
    int n_inner = 0;
    float glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int sl = 0; sl < 8; ++sl) {
      const int c = cand_in[sl];
      if (c < 0) { w[24 + sl] = 0x7f000000u; continue; }
      const Ref& r = cand[c];
      if (r.count == 0) { imask |= 1u << sl; ++n_inner; }
      else {
        if (r.count > 15) { ok = false; return; }
        lmask |= 1u << sl;
        counts |= (uint32_t)r.count << (4 * sl);
        for (int a = 0; a < 3; ++a) { glo[a] = std::min(glo[a], r.lo[a]); ghi[a] = std::max(ghi[a], r.hi[a]); }
      }
      {
        const int so = ref_solid(r);
        w[20 + (sl >> 1)] |= (uint32_t)((so >= 0 && so < 0xffff) ? so : 0xffff) << (16 * (sl & 1));
        w[24 + sl] = (so >= 0 && so < 0xffff) ? cone_word(r) : 0x7f000000u;
      }
      for (int a = 0; a < 3; ++a) {
        const double ql = std::floor(((double)r.lo[a] - (double)lo[a]) / scale[a]);
        const double qh = std::ceil(((double)r.hi[a] - (double)lo[a]) / scale[a]);
        const uint32_t bl = (uint32_t)std::max(0.0, std::min(255.0, ql)), bh = (uint32_t)std::max(0.0, std::min(255.0, qh));
        if (qh > 255.0) { ok = false; return; }                // (cannot happen: 255 scale >= extent)
        w[8 + 2 * a + (sl >> 2)] |= bl << (8 * (sl & 3));
        w[14 + 2 * a + (sl >> 2)] |= bh << (8 * (sl & 3));
      }
    }
    {
      int total = 0;
      for (int sl = 0; sl < 8; ++sl) total += (int)((counts >> (4 * sl)) & 15u);
      if (total > 64) { ok = false; return; }                  // (the kernel's candidate mask)
    }
    w[4] = child_base;
    w[5] = leaf_base;
    w[6] = imask | (lmask << 8);
    w[7] = counts;
    std::memcpy(&nodes[index * kWideWords], w, sizeof w);
    // leaf records of this node, slot order
    const float gc[3] = {0.5f * glo[0] + 0.5f * ghi[0], 0.5f * glo[1] + 0.5f * ghi[1], 0.5f * glo[2] + 0.5f * ghi[2]};
    for (int sl = 0; sl < 8; ++sl) {
      const int c = cand_in[sl];
      if (c < 0 || cand[c].count == 0) continue;
      for (int k = 0; k < cand[c].count; ++k) {
        leaf_prim.push_back(order[(size_t)cand[c].child + k]);
        leaf_center.insert(leaf_center.end(), gc, gc + 3);
      }
    }
    // inner children: consecutive nodes, slot order
    nodes.resize(nodes.size() + (size_t)n_inner * kWideWords, 0u);
    int rank = 0;
    for (int sl = 0; sl < 8; ++sl) {
      const int c = cand_in[sl];
      if (c < 0 || cand[c].count != 0) continue;
      const int child = cand[c].child;
      fill((size_t)child_base + rank, child, d + 1);
      if (!ok) return;
      ++rank;
    }
  }
};


// ---- rectilinear grid for big analytic scenes (odw_grid.hip) ---------------------------------
// Planes per axis: one in the middle of every gap between the primitives' boxes (projected on the
// axis) -- a Draft array gets one element per cell --, then slabs wider than twice the width an
// even division into ~cbrt(n) cells per axis would give are cut evenly.  Cell lists (CSR): every
// primitive whose box touches the cell.  The walk is exact whatever the planes are; they only
// decide how many cells a ray crosses and how many primitives it tests per cell.
constexpr int kGridMaxAxis = 128;             // cells per axis (8 bits each in the walk's cell word)
constexpr uint32_t kGridMaxCellItems = 255;   // 8-bit count in the cell word
constexpr size_t kGridLdsBudget = 144 * 1024; // of the CU's 160 KB, one block per CU

int build_grid(odw_ctx* ctx, const std::vector<Box>& boxes, const std::vector<char>& dead) {
  DeviceGrid& G = ctx->P.grid;
  std::memset(&G, 0, sizeof G);
  static const bool off = getenv("ODW_NO_GRID") != nullptr;
  const int n = (int)boxes.size();
  std::vector<int> live;
  for (int p = 0; p < n; ++p)
    if (!dead[p]) live.push_back(p);
  if (off || live.empty()) return ODW_OK;
  Box all;
  all.reset();
  for (int p : live) all.grow(boxes[p]);
  double ext[3], vol = 1.0;
  for (int a = 0; a < 3; ++a) { ext[a] = std::max(all.hi[a] - all.lo[a], 1e-9); vol *= ext[a]; }
  const double per_len = std::cbrt((double)live.size() / vol);     // cells per unit length for ~1 primitive per cell
  std::vector<double> planes[3];
  for (int a = 0; a < 3; ++a) {
    std::vector<std::pair<double, double>> iv;
    for (int p : live) iv.emplace_back(boxes[p].lo[a], boxes[p].hi[a]);
    std::sort(iv.begin(), iv.end());
    const double pad = 1e-6 * (1.0 + ext[a]);
    std::vector<double> b{all.lo[a] - pad};
    double cover = iv[0].second;
    for (size_t k = 1; k < iv.size(); ++k) {
      if (iv[k].first > cover) b.push_back(0.5 * (cover + iv[k].first));
      cover = std::max(cover, iv[k].second);
    }
    b.push_back(all.hi[a] + pad);
    const double target = 1.0 / std::max(per_len, 1e-12);          // width of a cell of the even division
    std::vector<double> cut{b[0]};
    for (size_t k = 1; k < b.size(); ++k) {
      const double wdt = b[k] - b[k - 1];
      const int parts = wdt > 2.0 * target ? (int)std::min<double>(kGridMaxAxis, std::floor(wdt / target + 0.5)) : 1;
      for (int j = 1; j <= parts; ++j) cut.push_back(j == parts ? b[k] : b[k - 1] + wdt * j / parts);
    }
    if ((int)cut.size() - 1 > kGridMaxAxis) {                     // too fine: even division
      cut.clear();
      for (int j = 0; j <= kGridMaxAxis; ++j) cut.push_back(b.front() + (b.back() - b.front()) * j / kGridMaxAxis);
      cut.back() = b.back();
    }
    planes[a] = cut;
  }
  const int nx = (int)planes[0].size() - 1, ny = (int)planes[1].size() - 1, nz = (int)planes[2].size() - 1;
  const size_t ncell = (size_t)nx * ny * nz;
  if (ncell > (1u << 21)) return ODW_OK;
  // cell ranges of every primitive (closed boxes: a box that ends on a plane is listed on both sides)
  auto range = [&](int a, double lo, double hi, int& i0, int& i1) {
    const std::vector<double>& b = planes[a];
    const int m = (int)b.size() - 1;
    i0 = (int)(std::upper_bound(b.begin(), b.end(), lo) - b.begin()) - 1;     // last plane <= lo
    if (i0 > 0 && b[i0] == lo) --i0;
    i1 = (int)(std::lower_bound(b.begin(), b.end(), hi) - b.begin()) - 1;     // slab whose upper plane >= hi
    if (i1 + 1 < m && b[i1 + 1] == hi) ++i1;
    i0 = std::max(0, std::min(m - 1, i0));
    i1 = std::max(i0, std::min(m - 1, i1));
  };
  std::vector<uint32_t> count(ncell, 0);
  std::vector<int> r(6 * (size_t)live.size());
  for (size_t k = 0; k < live.size(); ++k) {
    const Box& bx = boxes[live[k]];
    int* q = &r[6 * k];
    range(0, bx.lo[0], bx.hi[0], q[0], q[1]);
    range(1, bx.lo[1], bx.hi[1], q[2], q[3]);
    range(2, bx.lo[2], bx.hi[2], q[4], q[5]);
    for (int z = q[4]; z <= q[5]; ++z)
      for (int y = q[2]; y <= q[3]; ++y)
        for (int x = q[0]; x <= q[1]; ++x) ++count[x + (size_t)nx * (y + (size_t)ny * z)];
  }
  size_t total = 0;
  std::vector<uint32_t> first(ncell);
  for (size_t c = 0; c < ncell; ++c) {
    if (count[c] > kGridMaxCellItems) return ODW_OK;              // crowded beyond the cell word: BVH kernels
    first[c] = (uint32_t)total;
    total += count[c];
  }
  if (total >= (1u << 24)) return ODW_OK;
  std::vector<uint32_t> item_prim(std::max<size_t>(total, 1)), fill(ncell, 0);
  for (size_t k = 0; k < live.size(); ++k) {
    const int* q = &r[6 * k];
    for (int z = q[4]; z <= q[5]; ++z)
      for (int y = q[2]; y <= q[3]; ++y)
        for (int x = q[0]; x <= q[1]; ++x) {
          const size_t c = x + (size_t)nx * (y + (size_t)ny * z);
          item_prim[first[c] + fill[c]++] = (uint32_t)live[k];
        }
  }
  std::vector<uint32_t> cells(ncell);
  for (size_t c = 0; c < ncell; ++c) cells[c] = first[c] | (count[c] << 24);
  bool spheres = true;
  for (int p : live) {
    const int32_t* pi = &ctx->h_prim_i32[4 * (size_t)p];
    if (pi[0] != ODW_PRIM_SPHERE || ((pi[3] >> 24) & 0xff) != 0) { spheres = false; break; }
  }
  std::vector<double> bounds;
  for (int a = 0; a < 3; ++a) bounds.insert(bounds.end(), planes[a].begin(), planes[a].end());
  int rc;
  if ((rc = upload(ctx, ctx->grid_bounds, bounds.data(), bounds.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->grid_cells, cells.data(), cells.size() * sizeof(uint32_t)))) return rc;
  size_t item_bytes;
  std::vector<double> recs;
  if (spheres) {
    // 48-byte records: centre (global; prim_f64 12..15 = R, cx, cy, cz as the flat kernel reads them),
    // radius, {primitive, group | solid << 8}, the primitive's flag word
    recs.resize(std::max<size_t>(total, 1) * 6, 0.0);
    for (size_t k = 0; k < total; ++k) {
      const uint32_t p = item_prim[k];
      const double* par = ctx->h_prim_f64.data() + 16 * (size_t)p + 12;
      const int32_t* pi = &ctx->h_prim_i32[4 * (size_t)p];
      double* o = &recs[6 * k];
      o[0] = par[1]; o[1] = par[2]; o[2] = par[3]; o[3] = par[0];
      const uint64_t bits = (uint64_t)p | ((uint64_t)(uint32_t)((pi[1] & 0xff) | ((pi[2] >> ODW_SOLID_SHIFT) << 8)) << 32);
      std::memcpy(&o[4], &bits, sizeof bits);
      const uint64_t flag_word = (uint64_t)(uint32_t)pi[2];          // (flags | facemask << 8 | solid << 16, for the interaction)
      std::memcpy(&o[5], &flag_word, sizeof flag_word);
    }
    item_bytes = recs.size() * sizeof(double);
    if ((rc = upload(ctx, ctx->grid_items, recs.data(), item_bytes))) return rc;
  } else {
    item_prim.resize((item_prim.size() + 1) & ~(size_t)1, 0u);      // whole doubles (the LDS copy moves 8 bytes at a time)
    item_bytes = item_prim.size() * sizeof(uint32_t);
    if ((rc = upload(ctx, ctx->grid_items, item_prim.data(), item_bytes))) return rc;
  }
  if ((rc = upload_done(ctx))) return rc;                           // host vectors die with this scope
  // the kernel's LDS image (odw_grid_kernel, same arithmetic): planes | per-wave words | ray rings | cells | items
  const size_t nbp = bounds.size();
  const size_t word_off = 2 * nbp;
  const size_t ring_off = (word_off + (size_t)ODW_GRID_WAVES * ODW_GRID_WAVE_WORDS + 1) / 2;
  const size_t cell_off = 2 * (ring_off + (size_t)ODW_GRID_WAVES * ODW_GRID_RING_DOUBLES);
  const size_t fixed = cell_off * sizeof(uint32_t);
  const size_t staged = (((cell_off + ncell + 3) & ~(size_t)3) / 2) * sizeof(double) + item_bytes;
  G.bounds = (const double*)ctx->grid_bounds.p;
  G.cells = (const uint32_t*)ctx->grid_cells.p;
  G.items = ctx->grid_items.p;
  G.nx = nx; G.ny = ny; G.nz = nz;
  G.n_items = (int32_t)total;
  G.spheres = spheres ? 1 : 0;
  G.in_lds = staged + 16 <= kGridLdsBudget ? 1 : 0;
  G.lds_bytes = (uint32_t)((G.in_lds ? staged : fixed) + 16);
  return ODW_OK;
}

// the primitives' boxes and 64-byte headers (host only: ctx->h_prim_hdr, ctx->h_dead)
void compute_boxes(odw_ctx* ctx, std::vector<Box>& boxes, std::vector<char>& dead) {
  const int n = ctx->P.scene.n_prims;
  // boxes contain every point the tolerance rules may accept
  const double slack = 2.0 * (ctx->have_limits ? ctx->P.lim.dist_tol : 1e-2);
  boxes.assign(n, Box());
  std::vector<double>& flat = ctx->h_prim_hdr;
  flat.assign((size_t)std::max(1, n) * 8, 0.0);   // 64-byte headers
  for (int p = 0; p < n; ++p)
    boxes[p] = world_box(ctx->h_prim_f64.data() + 16 * (size_t)p, ctx->h_prim_i32[4 * p], slack);
  // A face that exists only inside other primitives (operands of a Common, the base of a Cut for
  // its tool) lies in their boxes too: the box of a lens cap is the lens, not the sphere.
  // Primitives without faces (pure operands) and faces that cannot exist get a box no ray meets.
  std::vector<Box> full = boxes;
  dead.assign(n, 0);
  for (int p = 0; p < n; ++p) {
    const int cw = ctx->h_prim_i32[4 * p + 3], off = cw & 0xffffff, cnt = (cw >> 24) & 0xff;
    for (int c = off; c < off + cnt && c < (int)ctx->h_cond.size(); ++c) {
      if (ctx->h_cond[c] >= 0) continue;                       // must be OUTSIDE that one: no bound
      const Box& o = full[ctx->h_cond[c] & 0x7fffffff];
      for (int a = 0; a < 3; ++a) {
        boxes[p].lo[a] = std::max(boxes[p].lo[a], o.lo[a]);
        boxes[p].hi[a] = std::min(boxes[p].hi[a], o.hi[a]);
      }
    }
    const int facemask = (ctx->h_prim_i32[4 * p + 2] >> ODW_FACEMASK_SHIFT) & 0xff;
    dead[p] = facemask == 0 || boxes[p].lo[0] > boxes[p].hi[0] || boxes[p].lo[1] > boxes[p].hi[1] ||
              boxes[p].lo[2] > boxes[p].hi[2];
    if (dead[p])
      for (int a = 0; a < 3; ++a) boxes[p].lo[a] = boxes[p].hi[a] = 1e30;
  }
  // ODW_FLAG_ISOLATED (odw_device.h): solids whose box keeps clear of every other solid's
  {
    std::map<int, Box> solid_box;
    for (int p = 0; p < n; ++p) {
      ctx->h_prim_i32[4 * p + 2] &= ~ODW_FLAG_ISOLATED;
      if (dead[p]) continue;
      const int sid = ctx->h_prim_i32[4 * p + 2] >> ODW_SOLID_SHIFT;
      auto it = solid_box.find(sid);
      if (it == solid_box.end()) { solid_box[sid] = boxes[p]; continue; }
      for (int a = 0; a < 3; ++a) {
        it->second.lo[a] = std::min(it->second.lo[a], boxes[p].lo[a]);
        it->second.hi[a] = std::max(it->second.hi[a], boxes[p].hi[a]);
      }
    }
    const double gap = 2.0 * slack;                             // 4 distTol
    static const bool enabled = !(getenv("ODW_ISOLATED") && getenv("ODW_ISOLATED")[0] == '0');   // (A/B runs)
    if (enabled && solid_box.size() <= 64 && solid_box.count(0x7fff) == 0)  // (0x7fff: solid ids that did not fit the word)
      for (int p = 0; p < n; ++p) {
        if (dead[p]) continue;
        const int sid = ctx->h_prim_i32[4 * p + 2] >> ODW_SOLID_SHIFT;
        const Box& mine = solid_box[sid];
        bool alone = true;
        for (const auto& other : solid_box) {
          if (other.first == sid) continue;
          bool apart = false;
          for (int a = 0; a < 3; ++a)
            apart |= mine.lo[a] - other.second.hi[a] > gap || other.second.lo[a] - mine.hi[a] > gap;
          if (!apart) { alone = false; break; }
        }
        if (alone) ctx->h_prim_i32[4 * p + 2] |= ODW_FLAG_ISOLATED;
      }
  }
  for (int p = 0; p < n; ++p) {
    double* h = flat.data() + 8 * (size_t)p;
    for (int a = 0; a < 3; ++a) { h[a] = boxes[p].lo[a]; h[3 + a] = boxes[p].hi[a]; }
    std::memcpy(h + 6, &ctx->h_prim_i32[4 * (size_t)p], 4 * sizeof(int32_t));
  }
  ctx->h_dead = dead;
}

int build_bvh(odw_ctx* ctx) {
  const int n = ctx->P.scene.n_prims;
  ctx->P.scene.n_nodes = 0;
  ctx->bvh_dirty = false;
  ctx->spec_dirty = true;
  std::vector<Box> boxes;
  std::vector<char> dead;
  compute_boxes(ctx, boxes, dead);
  const std::vector<double>& flat = ctx->h_prim_hdr;
  {
    int rc = upload(ctx, ctx->prim_hdr, flat.data(), flat.size() * sizeof(double));
    // (compute_boxes has set ODW_FLAG_ISOLATED in the flag words)
    if (!rc && n > 0) rc = upload(ctx, ctx->prim_i32, ctx->h_prim_i32.data(), ctx->h_prim_i32.size() * sizeof(int32_t));
    if (rc) return rc;
    if ((rc = upload_done(ctx))) return rc;
    ctx->P.scene.prim_hdr = (const double*)ctx->prim_hdr.p;
  }
  const int bvh_threshold = ctx->flat_limit;
  bool has_triangles = false, has_paraboloids = false;
  for (int p = 0; p < n; ++p) {
    has_triangles |= ctx->h_prim_i32[4 * p] == ODW_PRIM_TRIANGLE;
    has_paraboloids |= ctx->h_prim_i32[4 * p] == ODW_PRIM_PARABOLOID;
  }
  std::memset(&ctx->P.grid, 0, sizeof ctx->P.grid);
  ctx->P.scene.bvh_leaf = nullptr;
  ctx->P.scene.bvh_wide = nullptr;
  // (triangles are only known to the BVH kernels, paraboloids to the BVH and grid kernels)
  if (n <= bvh_threshold && !has_triangles && !has_paraboloids) return ODW_OK;
  if (!has_triangles) {
    int rc = build_grid(ctx, boxes, dead);
    if (rc) return rc;
  }
  // float32 traversal boxes: enlarge by what float rounding of the ray origin
  // and of the slab arithmetic can cost (see ray_box_f32 in odw_kernels.hip)
  for (int p = 0; p < n; ++p)
    for (int a = 0; a < 3; ++a) {
      const double s = 1e-4 + 4e-7 * (std::fabs(boxes[p].lo[a]) + std::fabs(boxes[p].hi[a]));
      boxes[p].lo[a] -= s;
      boxes[p].hi[a] += s;
    }
  BvhBuilder b(boxes);
  std::vector<int> ids;
  ids.reserve(n);
  for (int i = 0; i < n; ++i)
    if (!dead[i]) ids.push_back(i);
  if (ids.empty() && n > 0) ids.push_back(0);   // (a far-away box: the tree needs one leaf)
  b.nodes.reserve((size_t)n);
  const BvhBuilder::Ref root = b.build(ids, 0);
  if (root.count > 0) {   // everything in one leaf: wrap it into a root node
    BvhNode nd;
    for (int k = 0; k < 3; ++k) {
      nd.lo0[k] = round_down(root.box.lo[k]); nd.hi0[k] = round_up(root.box.hi[k]);
      // the second child does not exist.  Its box must be one no ray meets: an inverted box
      // (lo = +inf, hi = -inf) passes the slab test for every ray (min = -inf, max = +inf on
      // each axis) and would send the traversal back to node 0 for ever; a point far away fails
      // it for every direction
      nd.lo1[k] = 3.0e38f; nd.hi1[k] = 3.0e38f;
    }
    nd.child0 = root.child; nd.count0 = root.count;
    nd.child1 = 0; nd.count1 = 0;
    b.nodes.insert(b.nodes.begin(), nd);
  }
  if (b.max_depth + 2 > ODW_BVH_STACK) return fail(ctx, ODW_ERR_UNSUPPORTED, "BVH deeper than the LDS stack");
  int rc;
  if ((rc = upload(ctx, ctx->bvh_nodes, b.nodes.data(), b.nodes.size() * sizeof(BvhNode)))) return rc;
  if ((rc = upload(ctx, ctx->bvh_prims, b.order.data(), b.order.size() * sizeof(int)))) return rc;
  // the mesh kernel's eight-wide tree and leaf records (odw_mesh.hip: ODW_LEAF_WORDS): the facet relative to the centre
  // of the leaf group of its node, in float32, with the bounds the conservative filter needs
  ctx->P.scene.bvh_leaf = nullptr;
  ctx->P.scene.bvh_wide = nullptr;
  std::vector<float> recs;
  // (read at every build: A/B runs and the test that holds the two kernels against each other)
  const bool mesh_kernel = !(getenv("ODW_MESH_KERNEL") && getenv("ODW_MESH_KERNEL")[0] == '0');
  std::vector<int> prim_solid((size_t)n);
  for (int p = 0; p < n; ++p) prim_solid[p] = ctx->h_prim_i32[4 * (size_t)p + 2] >> ODW_SOLID_SHIFT;
  WideBvh wide(b.nodes, b.order, prim_solid);
  std::vector<float> out_normal;
  if (has_triangles && mesh_kernel) {
    // normal cones for rays inside STRICTLY convex tessellated solids (ODW_FLAG_STRICTLY_CONVEX; node words 24..31;
    // ODW_MESH_CONES=0: none).  The margin: a ray that starts on a facet whose edges are all closed is out of that facet's
    // area by 1e-9 of its edges at most; every point of a facet lies on or below the plane of every other facet up to
    // rounding (what the flag says: 1e-13 of the mesh's size per edge, taken a hundred times wider here); the point itself
    // is rounded (~1e-13 of the coordinates): above a dropped facet's plane by less than `above`, met at t < above / margin.
    const bool cones_off = getenv("ODW_MESH_CONES") && getenv("ODW_MESH_CONES")[0] == '0';      // (read at every build, as ODW_MESH_KERNEL)
    double size = 0.0, reach = 0.0;
    out_normal.assign(3 * (size_t)n, std::numeric_limits<float>::quiet_NaN());
    bool any = false;
    for (int p = 0; p < n && !cones_off; ++p) {
      const int32_t* pi = &ctx->h_prim_i32[4 * (size_t)p];
      if (pi[0] != ODW_PRIM_TRIANGLE || !(pi[2] & ODW_FLAG_CONVEX) || !(pi[2] & ODW_FLAG_STRICTLY_CONVEX)) continue;
      const double* pf = ctx->h_prim_f64.data() + 16 * (size_t)p;
      const double sg = (pi[2] & ODW_FLAG_FLIP_NORMAL) ? -1.0 : 1.0;
      for (int a = 0; a < 3; ++a) {
        out_normal[3 * (size_t)p + a] = (float)(sg * pf[9 + a]);
        size = std::max(size, std::fabs(pf[3 + a]) + std::fabs(pf[6 + a]));
        reach = std::max(reach, std::max(std::fabs(boxes[p].lo[a]), std::fabs(boxes[p].hi[a])));
      }
      any = true;
    }
    if (any) {
      // (size: the longest facet edge, and more; the mesh is at most the extent of all such facets together: reach both ways)
      const double above = 1e-9 * size + 1e-11 * 2.0 * reach + 1e-12 * reach;
      wide.margin = std::max(0.02, 2.0 * above / std::max(ctx->P.lim.dist_tol, 1e-300));
      wide.out_normal = out_normal.data();
    }
    wide.build();
    if (wide.ok) {
      recs.assign(std::max<size_t>(wide.leaf_prim.size(), 1) * ODW_LEAF_WORDS, 0.0f);
      for (size_t j = 0; j < wide.leaf_prim.size(); ++j) {
        const int p = wide.leaf_prim[j];
        float* r = &recs[j * ODW_LEAF_WORDS];
        const float* c = &wide.leaf_center[3 * j];
        const double* pf = ctx->h_prim_f64.data() + 16 * (size_t)p;
        const int32_t* pi = &ctx->h_prim_i32[4 * (size_t)p];
        uint32_t gs = (uint32_t)(pi[1] & 0xff) | ((uint32_t)((pi[2] >> ODW_SOLID_SHIFT) & 0x7fff) << 8);
        float smax = 0.0f, err = 0.0f;
        if (pi[0] == ODW_PRIM_TRIANGLE) {
          double l1[2] = {0.0, 0.0};
          float e1[3], e2[3];
          for (int a = 0; a < 3; ++a) {
            r[a] = (float)(pf[a] - (double)c[a]);
            e1[a] = (float)pf[3 + a];
            e2[a] = (float)pf[6 + a];
            l1[0] += std::fabs(pf[3 + a]);
            l1[1] += std::fabs(pf[6 + a]);
          }
          r[3] = e1[0]; r[4] = e1[1]; r[5] = e1[2]; r[6] = e2[0]; r[7] = e2[1]; r[8] = e2[2];
          smax = round_up(std::max(0.0, std::max(pf[12], std::max(pf[13], pf[14]))));
          err = round_up(4e-6 * std::max(l1[0], l1[1]));
        } else {
          gs |= 0x80000000u;
        }
        std::memcpy(&r[9], &gs, 4);
        r[10] = smax;
        r[11] = err;
        std::memcpy(&r[12], &p, 4);
        r[13] = c[0]; r[14] = c[1]; r[15] = c[2];
      }
      if (getenv("ODW_MESH_CONE_STATS")) {             // (diagnostics: how many slots carry a cone)
        size_t slots = 0, cones = 0;
        for (size_t k = 0; k + kWideWords <= wide.nodes.size(); k += kWideWords)
          for (int sl = 0; sl < 8; ++sl)
            if ((wide.nodes[k + 6] | (wide.nodes[k + 6] >> 8)) & (1u << sl)) { ++slots; cones += (wide.nodes[k + 24 + sl] >> 24) != 127u; }
        fprintf(stderr, "[odw mesh cones] margin %.4g, %zu of %zu slots carry a cone\n", wide.margin, cones, slots);
      }
      if ((rc = upload(ctx, ctx->bvh_leaf, recs.data(), recs.size() * sizeof(float)))) return rc;
      if ((rc = upload(ctx, ctx->bvh_wide, wide.nodes.data(), wide.nodes.size() * sizeof(uint32_t)))) return rc;
      for (int a = 0; a < 3; ++a) {       // node 0 as the kernel decodes it: corner + 255 units of its scale
        float corner, unit;
        const uint32_t eb = ((wide.nodes[3] >> (8 * a)) & 0xffu) << 23;
        std::memcpy(&corner, &wide.nodes[a], 4);
        std::memcpy(&unit, &eb, 4);
        ctx->P.scene.wide_lo[a] = (double)corner;
        ctx->P.scene.wide_hi[a] = (double)corner + 255.0 * (double)unit;
      }
    }
  }
  { int rc_ = upload_done(ctx); if (rc_) return rc_; }  // host vectors die with this scope
  ctx->P.scene.bvh_nodes = (const float*)ctx->bvh_nodes.p;
  ctx->P.scene.bvh_prims = (const int32_t*)ctx->bvh_prims.p;
  ctx->P.scene.bvh_leaf = recs.empty() ? nullptr : (const float*)ctx->bvh_leaf.p;
  ctx->P.scene.bvh_wide = recs.empty() ? nullptr : (const uint32_t*)ctx->bvh_wide.p;
  ctx->P.scene.n_nodes = (int)b.nodes.size();
  return ODW_OK;
}

// ---- device-side ordering of the hit list (odw_fetch_hits) -----------------
__global__ void hit_keys_kernel(const odw_hit* __restrict__ hits, uint64_t n, uint64_t sentinel, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ vals) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t tag = hits[i].tag;
    keys[i] = tag == ODW_TAG_UNUSED ? sentinel : ODW_HIT_RAY(tag);   // unused slots sort behind every ray
    vals[i] = (uint32_t)i;
  }
}

__global__ void seg_keys_kernel(const odw_segment* __restrict__ segs, uint64_t n, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ vals) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t tag = segs[i].tag;
    keys[i] = (ODW_SEG_RAY(tag) << 12) | ODW_SEG_ORDINAL(tag);    // 52 bits
    vals[i] = (uint32_t)i;
  }
}

// four lanes move one 64-byte row (16 B each): coalesced reads of the index
// list, 64-B gathers, fully coalesced writes
__global__ void hit_gather_kernel(const odw_hit* __restrict__ hits, const uint32_t* __restrict__ order,
                                  uint64_t n, odw_hit* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = t >> 2;
  if (row < n) {
    const double2* src = reinterpret_cast<const double2*>(hits + order[row]);
    reinterpret_cast<double2*>(out + row)[t & 3] = src[t & 3];
  }
}

}  // namespace
#include "odw_spec.hip"
namespace {

// Mesh launches of device-generated rays: the order the rays are handed out in = sorted by where they start and where
// they point (odw_mesh.hip: odw_ray_key_kernel), so that the 64 rays of a wave are neighbours among all rays of the
// launch -- the same nodes, the same leaves, the same cache lines; a ray's rows depend on its number only, so the
// results are those of the unsorted launch.  Cost: one generation pass for the keys + a radix sort of (key, number)
// pairs (1e7 rays: ~0.8 ms against 10 - 17 ms of tracing); short launches and ODW_MESH_PRESORT=0 keep the plain order.
int presort_rays(odw_ctx* ctx, uint64_t first, uint64_t n, uint64_t seed) {
  // (read at every launch: A/B runs and the test that holds the two orders against each other)
  const bool off = getenv("ODW_MESH_PRESORT") && getenv("ODW_MESH_PRESORT")[0] == '0';
  const char* e_min = getenv("ODW_MESH_PRESORT_MIN");
  const uint64_t min_rays = e_min ? (uint64_t)atoll(e_min) : (1ull << 16);
  if (off || n < min_rays || n > 0x7FFFFFFFull) return ODW_OK;
  int rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = ensure(ctx, ctx->sort_keys[k], n * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals[k], n * sizeof(uint32_t)))) return rc;
  }
  const DeviceScene& sc = ctx->P.scene;
  double lo[3], scale[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = sc.wide_lo[a];
    const double w = sc.wide_hi[a] - sc.wide_lo[a];
    scale[a] = w > 0 ? 1024.0 / w : 0.0;
  }
  uint64_t* k_in = (uint64_t*)ctx->sort_keys[0].p;
  uint64_t* k_out = (uint64_t*)ctx->sort_keys[1].p;
  uint32_t* v_in = (uint32_t*)ctx->sort_vals[0].p;
  uint32_t* v_out = (uint32_t*)ctx->sort_vals[1].p;
  // a point source with focal length 0 starts every ray at one point: the direction bits alone, as 32-bit keys
  // (four passes of eight bits over half the bytes)
  const bool dir_only = ctx->h_source.finite_focal && ctx->h_source.focal_length == 0.0;
  const unsigned kgrid = (unsigned)((n + 255) / 256);
  size_t tmp_bytes = 0;
  if (dir_only) {
    uint32_t* k32_in = (uint32_t*)k_in;
    uint32_t* k32_out = (uint32_t*)k_out;
    hipLaunchKernelGGL(odw_ray_key_kernel<uint32_t>, dim3(kgrid), dim3(256), 0, ctx->stream, ctx->P.source, first, n, seed,
                       lo[0], lo[1], lo[2], scale[0], scale[1], scale[2], k32_in, v_in);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k32_in, k32_out, v_in, v_out, (int)n, 0, 32, ctx->stream));
    if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k32_in, k32_out, v_in, v_out, (int)n, 0, 32, ctx->stream));
  } else {
    hipLaunchKernelGGL(odw_ray_key_kernel<uint64_t>, dim3(kgrid), dim3(256), 0, ctx->stream, ctx->P.source, first, n, seed,
                       lo[0], lo[1], lo[2], scale[0], scale[1], scale[2], k_in, v_in);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, 62, ctx->stream));
    if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, 62, ctx->stream));
  }
  ctx->P.ray_order = v_out;
  ctx->ph_valid = false;           // (the sort buffers are shared with odw_hits_select)
  return ODW_OK;
}

#if ODW_GRID_SORTED
// Grid launches of device-generated rays (diagnostic builds, -DODW_GRID_SORTED=1): hand-out order = sorted by the top `bits`
// bits of the Morton key of the two uniform numbers a ray's direction is drawn from (odw_grid.hip: odw_ray_ukey_kernel;
// Philox only).  ODW_GRID_PRESORT = bits; the rows of a ray depend on its number only.  See profiles/r05/README.md.
int presort_rays_grid(odw_ctx* ctx, uint64_t first, uint64_t n, uint64_t seed, int bits) {
  if (bits <= 0 || n < (1ull << 16) || n > 0x7FFFFFFFull) return ODW_OK;
  bits = std::min(bits, 32);
  int rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = ensure(ctx, ctx->sort_keys[k], n * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals[k], n * sizeof(uint32_t)))) return rc;
  }
  uint32_t* k_in = (uint32_t*)ctx->sort_keys[0].p;
  uint32_t* k_out = (uint32_t*)ctx->sort_keys[1].p;
  uint32_t* v_in = (uint32_t*)ctx->sort_vals[0].p;
  uint32_t* v_out = (uint32_t*)ctx->sort_vals[1].p;
  const unsigned kgrid = (unsigned)((n + 255) / 256);
  size_t tmp_bytes = 0;
  hipLaunchKernelGGL(odw_ray_ukey_kernel, dim3(kgrid), dim3(256), 0, ctx->stream, first, n, seed, k_in, v_in);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 32 - bits, 32, ctx->stream));
  if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 32 - bits, 32, ctx->stream));
  ctx->P.ray_order = v_out;
  ctx->ph_valid = false;           // (the sort buffers are shared with odw_hits_select)
  return ODW_OK;
}
#endif

int launch_trace(odw_ctx* ctx, uint64_t first, uint64_t n, uint64_t seed, uint32_t flags,
                 const double* ray_o, const double* ray_d, const double* ray_p) {
  const bool explicit_rays = ray_o != nullptr;
  if (!ctx->have_scene || !ctx->have_limits) return fail(ctx, ODW_ERR_NO_SCENE, "scene/limits not uploaded");
  if (!explicit_rays && !ctx->have_source) return fail(ctx, ODW_ERR_NO_SCENE, "source not uploaded");
  if (n == 0) return ODW_OK;
  ctx->ph_valid = false;           // the hit list is about to change
  if (!ctx->batch_launch) {
    ctx->hit_ray_begin = ctx->hit_ray_end ? std::min<uint64_t>(ctx->hit_ray_begin, first) : first;
    ctx->hit_ray_end = std::max<uint64_t>(ctx->hit_ray_end, std::min<uint64_t>(first + n, 1ull << 48));
  }
  if (ctx->bvh_dirty) {
    int rc = build_bvh(ctx);
    if (rc) return rc;
  }
  if (ctx->spec_dirty) {
    // a scene kernel that cannot be built (no hiprtc on this machine, a compiler error) is not a reason to
    // stop tracing: the generic kernels run, odw_compile_scene / odw_last_error tell why
    if (spec_bind(ctx) != ODW_OK) ctx->spec_fn = nullptr;
  }
  if ((flags & ODW_TRACE_RECORD_HITS) && (ctx->batch_launch ? ctx->batch_seg_capacity : ctx->hit_capacity) == 0)
    return fail(ctx, ODW_ERR_CAPACITY, "ODW_TRACE_RECORD_HITS without odw_reserve_hits");
  if (flags & ODW_TRACE_RECORD_SEGMENTS) {
    if (ctx->seg_capacity == 0)
      return fail(ctx, ODW_ERR_CAPACITY, "ODW_TRACE_RECORD_SEGMENTS without odw_reserve_segments");
    if (first + n > ODW_SEG_MAX_RAY || ctx->P.lim.max_intersections > ODW_SEG_MAX_ORDINAL)
      return fail(ctx, ODW_ERR_INVALID, "ODW_TRACE_RECORD_SEGMENTS: ray index or max_intersections beyond the row tag");
  }
  if ((flags & ODW_TRACE_HISTOGRAM) && !ctx->P.det_enabled) flags &= ~ODW_TRACE_HISTOGRAM;
  TraceParams& P = ctx->P;
  const bool batch = ctx->batch_launch;
  std::memset(&P.batch, 0, sizeof P.batch);
  if (batch) {
    // scenes of one structure side by side: flat kernels only (the scene the context holds is scene 0 of the batch)
    if (P.scene.n_nodes || P.grid.nx > 0 || ctx->n_samplers > 0 || explicit_rays || (flags & ODW_TRACE_RECORD_SEGMENTS))
      return fail(ctx, ODW_ERR_UNSUPPORTED, "odw_trace_batch: batches are traced by the flat kernels (analytic scenes of up to 64 "
                                            "primitives, no stochastic surfaces, no segment rows)");
    flags &= ~(uint32_t)ODW_TRACE_HISTOGRAM;          // (one histogram cannot serve several scenes)
  }
  P.first_ray = first;
  P.n_rays = n;
  P.seed = seed;
  P.flags = flags;
  P.samplers = (const DeviceSurfaceSampler*)ctx->d_samplers.p;
  P.group_sampler = (const int32_t*)ctx->d_group_sampler.p;
  P.ray_origins = ray_o;
  P.ray_dirs = ray_d;
  P.ray_powers = ray_p;
  P.ray_stride = n;
  P.dbg = (unsigned long long*)ctx->dbg.p;
  P.out.hits = (odw_hit*)(batch ? ctx->batch_hits.p : ctx->hits.p);
  P.out.hit_capacity = batch ? ctx->batch_seg_slots : ctx->hit_slots;
  // block reservations need room for the unused slots they can leave behind: < 64 per block and the
  // last block of every wave of the grid
  P.out.hit_block = 0;
  P.out.hit_count = (unsigned long long*)(batch ? ctx->batch_hit_count.p : ctx->hit_count.p);
  P.out.hist = (unsigned long long*)ctx->hist.p;
  P.out.counters = (unsigned long long*)ctx->counters.p;
  P.out.chunk_counter = (unsigned long long*)ctx->chunk_counter.p;
  P.out.segs = (odw_segment*)ctx->segs.p;
  P.out.seg_capacity = ctx->seg_capacity;
  P.out.seg_count = (unsigned long long*)ctx->seg_count.p;
  P.out.row_of = (batch && ctx->batch_marked && (flags & ODW_TRACE_RECORD_HITS)) ? (uint32_t*)ctx->phb_row_of.p : nullptr;
  P.out.row_stride = batch ? (n + 31) / 32 * 32 : 0;
  P.out.pts = (P.out.row_of && ctx->batch_pts) ? (double*)ctx->phb_pts.p : nullptr;

  // persistent waves: one grid that fills the chip (4 blocks of 256 threads
  // per CU at 4 waves/SIMD, x2 so that a CU never waits for a block launch);
  // chunks of ODW_CHUNK rays are handed out dynamically inside the kernel
  static const int grid_mult = [] { const char* e = getenv("ODW_GRID_MULT"); int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  // big analytic scenes: grid kernel (no stochastic surfaces, no segment rows: those stay with the BVH kernels)
  // a scene compiled against its structure (odw_spec.hip): its own kernel, whatever else was built for it
  if (batch && ctx->spec_fn && !ctx->spec_batch_fn && !ctx->spec_batch_failed) {
    // the compiled kernel's BATCH variant: bound on the first batch launch of the structure (a compilation of its own,
    // cached like the other; odw_compile_scene's mode decides, as for single launches).  A variant that cannot be built
    // is not tried again for this binding, and its failure does not become the error of a launch that succeeds on the
    // generic kernel
    const std::string keep_err = ctx->err;
    if (spec_bind(ctx, true) != ODW_OK) {
      ctx->spec_batch_fn = nullptr;
      ctx->spec_batch_failed = true;
      ctx->err = keep_err;
    }
  }
  const bool use_spec = (batch ? ctx->spec_batch_fn != nullptr : ctx->spec_fn != nullptr) && ctx->spec_lean == ctx->lean &&
                        ctx->spec_stoch == (ctx->n_samplers > 0) && !(flags & ODW_TRACE_RECORD_SEGMENTS);
  const bool use_grid = !use_spec && P.grid.nx > 0 && ctx->n_samplers == 0 && !(flags & ODW_TRACE_RECORD_SEGMENTS);
  // Rays per hand-out unit.  A launch should hold many chunks per resident wave: with about one each -- 1e7 rays in
  // chunks of 2048 on 4096 resident waves -- the waves that get a second one set the launch's length.  Measured
  // (kernel ms at 1e7 / 1e8 rays): flat kernels 2048: 1.46 / 11.07, 1024: 1.44 / 10.93, 512: 1.39 / 10.97, 256: 1.43;
  // the ring kernels (grid, mesh: a ring fill is 64 rays whatever the chunk) 2048: 2.24 / 21.34, 512: 2.05 / 20.95,
  // 256: 1.96 / 20.86, and the mesh kernel at 1e7 rays and 6.5e4 facets 2048: 14.6, 256: 12.5, 64: 12.3.
  {
    static const uint64_t forced = [] { const char* e = getenv("ODW_CHUNK_RAYS"); return e ? (uint64_t)atoll(e) : 0ull; }();   // (A/B runs)
    const bool ring = use_grid || (!use_spec && P.scene.n_nodes && P.scene.bvh_leaf && !(flags & ODW_TRACE_RECORD_SEGMENTS));
    const uint64_t waves = (uint64_t)ctx->n_cu * 16;
    const uint64_t want = ring ? std::max<uint64_t>(64, std::min<uint64_t>(256, n / (waves * 32)))
                               : std::max<uint64_t>(512, std::min<uint64_t>(1024, n / (waves * 4)));
    P.chunk = (uint32_t)((forced ? std::max<uint64_t>(64, std::min<uint64_t>(ODW_CHUNK, forced)) : want) & ~(uint64_t)63);
  }
  uint64_t n_chunks = (n + P.chunk - 1) / P.chunk;
  if (batch) {
    // n = the rays of ONE scene; the launch hands out chunks_per_scene units per scene
    const uint64_t total = n * (uint64_t)ctx->batch_traced, waves = (uint64_t)ctx->n_cu * 16;
    static const uint64_t forced = [] { const char* e = getenv("ODW_CHUNK_RAYS"); return e ? (uint64_t)atoll(e) : 0ull; }();
    P.chunk = (uint32_t)((forced ? std::max<uint64_t>(64, std::min<uint64_t>(ODW_CHUNK, forced))
                                 : std::max<uint64_t>(512, std::min<uint64_t>(1024, total / (waves * 4)))) & ~(uint64_t)63);
    const uint64_t cps = (n + P.chunk - 1) / P.chunk;
    if (cps * (uint64_t)ctx->batch_traced >= (1ull << 32)) return fail(ctx, ODW_ERR_INVALID, "odw_trace_batch: too many hand-out units");
    P.batch.rays = n;
    P.batch.stride = ctx->batch_stride;
    P.batch.chunks_per_scene = (uint32_t)cps;
    P.batch.n_scenes = (uint32_t)ctx->batch_traced;
    n_chunks = cps * (uint64_t)ctx->batch_traced;
    // the batch's rays once, for all its scenes (from three scenes on: the pass writes 24 bytes per ray -- 48 where the
    // origins differ -- and every scene reads them; ODW_BATCH_SHARED_RAYS=0: every scene generates its own, =1: always)
    P.batch.gen_dirs = P.batch.gen_origins = nullptr;
    P.batch.gen_stride = 0;
    static const int shared_mode = [] { const char* e = getenv("ODW_BATCH_SHARED_RAYS"); return e ? atoi(e) : -1; }();
    if (shared_mode != 0 && (ctx->batch_traced >= 3 || shared_mode == 1)) {
      const bool one_origin = ctx->h_source.finite_focal && ctx->h_source.focal_length == 0.0;
      const uint64_t gs = (n + 31) / 32 * 32;
      const size_t bytes = (size_t)(gs * (one_origin ? 3 : 6) + 4) * sizeof(double);
      // (sized by odw_batch_reserve before a sweep; here only if nobody did: a launch without the pass is still right)
      if (ctx->batch_rays_buf.bytes >= bytes || ensure(ctx, ctx->batch_rays_buf, bytes) == ODW_OK) {
        double* dirs = (double*)ctx->batch_rays_buf.p;
        double* orgs = one_origin ? nullptr : dirs + 3 * gs + 4;
        P.batch.gen_dirs = dirs;                    // (the pass itself: below, inside the launch's timed interval)
        P.batch.gen_origins = orgs;
        P.batch.gen_stride = gs;
      } else {
        ctx->err.clear();
      }
    }
  }
  // Batch launches share the GPU with the post-hoc chains of other contexts (a sweep keeps several groups in flight): three
  // blocks per CU instead of all four leave a quarter of every SIMD's registers to their kernels and to the runtime's copy
  // kernels, which otherwise wait until a persistent block retires, i.e. for the whole launch (ODW_BATCH_GRID_MULT; measured
  // on the 64 x 1e7 sweep with 16 hardware queues: 98 ms -> 86 ms per sweep)
  static const int batch_mult = [] { const char* e = getenv("ODW_BATCH_GRID_MULT"); int v = e ? atoi(e) : 0; return v > 0 ? v : 3; }();
  const uint64_t cap = (uint64_t)ctx->n_cu * (batch ? std::min(batch_mult, grid_mult) : grid_mult);
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_chunks + 3) / 4, cap));
  const uint64_t grid_blocks = std::max<uint64_t>(1, std::min<uint64_t>((n_chunks + ODW_GRID_WAVES - 1) / ODW_GRID_WAVES, (uint64_t)ctx->n_cu));
  // scenes with facets: the mesh kernel (same exclusions)
  const bool use_mesh = !use_spec && !use_grid && P.scene.n_nodes && P.scene.bvh_leaf && !(flags & ODW_TRACE_RECORD_SEGMENTS);
  const uint64_t n_waves = use_grid ? grid_blocks * ODW_GRID_WAVES : (uint64_t)grid * 4;
  if ((!P.scene.n_nodes || use_grid || use_spec || use_mesh) && !ctx->swapping)   // flat, grid and mesh kernels only (see record_hit)
    for (uint32_t b = kHitBlock; b >= 128 && b >= kHitBlock / 4 && !P.out.hit_block; b /= 2)   // (a short list: smaller blocks before none)
      if (batch ? ctx->batch_seg_slots >= ctx->batch_seg_capacity + hit_block_room(ctx->batch_seg_capacity, n_waves, b)
                : ctx->hit_slots >= ctx->hit_capacity + hit_block_room(ctx->hit_capacity, n_waves, b)) P.out.hit_block = b;
  HIPCHK(ctx, hipMemsetAsync(ctx->chunk_counter.p, 0, sizeof(uint64_t), ctx->stream));
  const size_t lds = P.scene.n_nodes ? (size_t)ODW_BVH_STACK * 256 * sizeof(int) : 0;

  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  if (ctx->timing) {
    if (!ctx->free_events.empty()) {
      ev = ctx->free_events.back();
      ctx->free_events.pop_back();
    } else {
      HIPCHK(ctx, hipEventCreate(&ev.first));
      HIPCHK(ctx, hipEventCreate(&ev.second));
    }
    HIPCHK(ctx, hipEventRecord(ev.first, ctx->stream));
  }
  if (batch && P.batch.gen_dirs) {
    const unsigned gb = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ctx->n_cu * 16);
    hipLaunchKernelGGL(odw_batch_rays_kernel, dim3(gb), dim3(256), 0, ctx->stream, P.source, first, n, seed,
                       const_cast<double*>(P.batch.gen_dirs), const_cast<double*>(P.batch.gen_origins), P.batch.gen_stride);
  }
  const bool stoch = ctx->n_samplers > 0;
  P.ray_order = nullptr;
  P.interact_min = 1;
  P.refill_min = 0;
  if (use_mesh && !explicit_rays) {
    int rc = presort_rays(ctx, first, n, seed);       // (inside the timed window: part of the launch's cost)
    if (rc) return rc;
  }
#if ODW_GRID_SORTED
  if (use_grid) {
    // (diagnostic builds only; read at every launch: A/B runs)
    const char* e_bits = getenv("ODW_GRID_PRESORT");
    const char* e_gate = getenv("ODW_GRID_GATE");
    const char* e_refill = getenv("ODW_GRID_REFILL");
    P.refill_min = e_refill ? (uint32_t)std::max(1, std::min(64, atoi(e_refill))) : 1u;
    if (!explicit_rays && e_bits) {
      int rc = presort_rays_grid(ctx, first, n, seed, atoi(e_bits));
      if (rc) return rc;
      if (P.ray_order && e_gate) P.interact_min = (uint32_t)std::max(1, std::min(64, atoi(e_gate)));
    }
  }
#endif
  if (use_spec) {
    int rc = spec_launch(ctx, grid, batch);
    if (rc) return rc;
  } else if (batch) {
    if (ctx->lean) hipLaunchKernelGGL((odw_trace_kernel<false, false, false, true, true>), dim3(grid), dim3(256), 0, ctx->stream, P);
    else hipLaunchKernelGGL((odw_trace_kernel<false, false, false, false, true>), dim3(grid), dim3(256), 0, ctx->stream, P);
  } else if (use_grid) {
    const dim3 gb((unsigned)grid_blocks);
    const size_t glds = P.grid.lds_bytes;
#define ODW_GRID_LAUNCH(S, L, O)                                                                                 \
    do {                                                                                                       \
      /* (once per device and instantiation: a second context on another GPU of the process needs its own) */  \
      static uint64_t attr_set = 0;                                                                            \
      const uint64_t dev_bit = 1ull << (ctx->device & 63);                                                     \
      if (!(attr_set & dev_bit)) {                                                                             \
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&odw_grid_kernel<S, L, O>),              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));   /* + 4.5 KB static */ \
        attr_set |= dev_bit;                                                                                   \
      }                                                                                                        \
      hipLaunchKernelGGL((odw_grid_kernel<S, L, O>), gb, dim3(ODW_GRID_THREADS), glds, ctx->stream, P);        \
    } while (0)
#if ODW_GRID_SORTED
    if (P.ray_order) {
      if (P.grid.spheres) { if (P.grid.in_lds) ODW_GRID_LAUNCH(true, true, true); else ODW_GRID_LAUNCH(true, false, true); }
      else { if (P.grid.in_lds) ODW_GRID_LAUNCH(false, true, true); else ODW_GRID_LAUNCH(false, false, true); }
    } else
#endif
    {
      if (P.grid.spheres) { if (P.grid.in_lds) ODW_GRID_LAUNCH(true, true, false); else ODW_GRID_LAUNCH(true, false, false); }
      else { if (P.grid.in_lds) ODW_GRID_LAUNCH(false, true, false); else ODW_GRID_LAUNCH(false, false, false); }
    }
#undef ODW_GRID_LAUNCH
  } else if (use_mesh) {
    const size_t mlds = (size_t)ODW_MESH_STACK * ODW_MESH_THREADS * 2 * sizeof(int) +
                        (size_t)ODW_MESH_BLOCK_WAVES * (ODW_MESH_WAVE_WORDS * sizeof(uint32_t) + ODW_MESH_RING_DOUBLES * sizeof(double));
    if (stoch) hipLaunchKernelGGL(odw_mesh_kernel<true>, dim3(grid), dim3(ODW_MESH_THREADS), mlds, ctx->stream, P);
    else hipLaunchKernelGGL(odw_mesh_kernel<false>, dim3(grid), dim3(ODW_MESH_THREADS), mlds, ctx->stream, P);
  } else if (flags & ODW_TRACE_RECORD_SEGMENTS) {
    if (P.scene.n_nodes) {
      if (stoch) hipLaunchKernelGGL((odw_trace_kernel<true, true, true>), dim3(grid), dim3(256), lds, ctx->stream, P);
      else hipLaunchKernelGGL((odw_trace_kernel<true, false, true>), dim3(grid), dim3(256), lds, ctx->stream, P);
    } else {
      if (stoch) hipLaunchKernelGGL((odw_trace_kernel<false, true, true>), dim3(grid), dim3(256), 0, ctx->stream, P);
      else hipLaunchKernelGGL((odw_trace_kernel<false, false, true>), dim3(grid), dim3(256), 0, ctx->stream, P);
    }
  } else if (P.scene.n_nodes) {
    if (stoch) hipLaunchKernelGGL((odw_trace_kernel<true, true, false>), dim3(grid), dim3(256), lds, ctx->stream, P);
    else hipLaunchKernelGGL((odw_trace_kernel<true, false, false>), dim3(grid), dim3(256), lds, ctx->stream, P);
  } else {
    if (stoch) hipLaunchKernelGGL((odw_trace_kernel<false, true, false>), dim3(grid), dim3(256), 0, ctx->stream, P);
    else if (ctx->lean) hipLaunchKernelGGL((odw_trace_kernel<false, false, false, true>), dim3(grid), dim3(256), 0, ctx->stream, P);
    else hipLaunchKernelGGL((odw_trace_kernel<false, false, false>), dim3(grid), dim3(256), 0, ctx->stream, P);
  }
  HIPCHK(ctx, hipGetLastError());
  if (ctx->timing) {
    HIPCHK(ctx, hipEventRecord(ev.second, ctx->stream));
    ctx->events.push_back(ev);
  }
  if (!use_spec) spec_note_launch(ctx, n);
  return ODW_OK;
}

constexpr uint64_t kEmitChunk = 1ull << 24;

// component_major: 3 x n for the trace kernels; else n x 3 (odw_generate_rays hands that to the caller)
int emit_rays(odw_ctx* ctx, uint64_t first, uint64_t n, uint64_t seed, bool component_major = true) {
  int rc;
  if ((rc = ensure(ctx, ctx->em_o, std::min<uint64_t>(n, kEmitChunk) * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->em_d, std::min<uint64_t>(n, kEmitChunk) * 3 * sizeof(double)))) return rc;
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, (uint64_t)ctx->n_cu * 16));
  hipLaunchKernelGGL(odw_emit_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->h_emitter, first, n, seed,
                     (double*)ctx->em_o.p, (double*)ctx->em_d.p, component_major ? n : (uint64_t)1,
                     component_major ? (uint64_t)1 : (uint64_t)3);
  HIPCHK(ctx, hipGetLastError());
  return ODW_OK;
}

}  // namespace

extern "C" {

int odw_abi_version(void) { return ODW_ABI_VERSION; }

const char* odw_last_error(const odw_ctx* ctx) { return ctx ? ctx->err.c_str() : g_error.c_str(); }

namespace {
int ensure_results(odw_ctx* ctx, uint64_t n_bins) {
  const size_t bins = n_bins > 2 ? (size_t)n_bins : 2;
  const size_t need = (kResultsHead + bins) * sizeof(uint64_t);
  if (!ctx->results.p || ctx->results.bytes < need) {
    DevBuf fresh;
    HIPCHK(ctx, hipMalloc(&fresh.p, need));
    fresh.bytes = need;
    if (ctx->results.p) {            // a launch may still be writing the old block; its counters move over
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(fresh.p, ctx->results.p, kResultsHead * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      release(ctx->results);
    } else {
      HIPCHK(ctx, hipMemsetAsync(fresh.p, 0, kResultsHead * sizeof(uint64_t), ctx->stream));
    }
    ctx->results = fresh;
  }
  ctx->counters.p = ctx->results.p;
  ctx->counters.bytes = ODW_CNT_COUNT * sizeof(uint64_t);
  ctx->hist.p = (uint64_t*)ctx->results.p + kResultsHead;
  ctx->hist.bytes = ctx->results.bytes - kResultsHead * sizeof(uint64_t);
  return ODW_OK;
}
}  // namespace

int odw_create(int device, odw_ctx** out) {
  if (!out) return fail(nullptr, ODW_ERR_INVALID, "odw_create: null out");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, ODW_ERR_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(nullptr, ODW_ERR_INVALID, "odw_create: bad device index");
  odw_ctx* ctx = new (std::nothrow) odw_ctx();
  if (!ctx) return fail(nullptr, ODW_ERR_DEVICE, "out of host memory");
  ctx->device = device;
  if (const char* e = getenv("ODW_BVH_THRESHOLD")) ctx->flat_limit = atoi(e);
  if (const char* e = getenv("ODW_SPEC_HOT_RAYS")) ctx->spec_hot_rays = (uint64_t)atof(e);
  std::memset(&ctx->P, 0, sizeof ctx->P);
  ctx->P.wavelength = 500.0;
  std::memset(&ctx->det_desc, 0, sizeof ctx->det_desc);
  std::memset(&ctx->h_source, 0, sizeof ctx->h_source);
  std::memset(&ctx->h_det, 0, sizeof ctx->h_det);
  std::memset(&ctx->h_emitter, 0, sizeof ctx->h_emitter);
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess) {
    fail(nullptr, ODW_ERR_DEVICE, std::string("odw_create: ") + hipGetErrorString(e));
    delete ctx;
    return ODW_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  if (getenv("ODW_GRID_STATS")) {               // diagnostic builds of the grid kernel report here (odw_destroy prints)
    if (ensure(ctx, ctx->dbg, 32 * sizeof(uint64_t)) == ODW_OK) (void)hipMemset(ctx->dbg.p, 0, 32 * sizeof(uint64_t));
  }
  int rc = ensure_results(ctx, 0);
  if (!rc) rc = ensure(ctx, ctx->hit_count, 2 * sizeof(uint64_t));
  if (!rc) rc = ensure(ctx, ctx->chunk_counter, sizeof(uint64_t));
  if (!rc) rc = ensure(ctx, ctx->seg_count, sizeof(uint64_t));
  if (rc) { g_error = ctx->err; odw_destroy(ctx); return rc; }
  (void)hipMemsetAsync(ctx->results.p, 0, ctx->results.bytes, ctx->stream);
  (void)hipMemsetAsync(ctx->hit_count.p, 0, ctx->hit_count.bytes, ctx->stream);
  (void)hipMemsetAsync(ctx->seg_count.p, 0, ctx->seg_count.bytes, ctx->stream);
  *out = ctx;
  return ODW_OK;
}

void odw_destroy(odw_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  batch_unselect(ctx);
  if (ctx->dbg.p) {
    uint64_t v[32] = {0};
    (void)hipMemcpy(v, ctx->dbg.p, sizeof v, hipMemcpyDeviceToHost);
    const char* names[8] = {"A ring fills", "B segment setup", "C cell steps", "D all", "D resolve", "D interact", "candidates", "passed"};
    for (int k = 0; k < 8; ++k)
      fprintf(stderr, "[odw grid stats] %-16s runs %12llu  lanes %14llu  (%.1f per run)\n", names[k], (unsigned long long)v[2 * k],
              (unsigned long long)v[2 * k + 1], v[2 * k] ? (double)v[2 * k + 1] / (double)v[2 * k] : 0.0);
    // (mesh kernel, ODW_MESH_STATS: clock ticks of s_memtime every wave spent in each phase, waits included)
    uint64_t total = 0;
    for (int k = 16; k < 24; ++k) total += v[k];
    const char* phases[8] = {"A refill", "B setup", "C walk", "D leaves", "-", "D interact", "-", "-"};
    if (total)
      for (int k = 0; k < 6; ++k)
        fprintf(stderr, "[odw mesh time] %-16s %14llu ticks  %5.1f %%\n", phases[k], (unsigned long long)v[16 + k], 100.0 * (double)v[16 + k] / (double)total);
    // (flat kernels, ODW_FLAT_STATS: refill + generation / nearest-hit search / interaction and recording)
    if (v[28] + v[29] + v[30])
      for (int k = 0; k < 3; ++k)
        fprintf(stderr, "[odw flat time] %-28s %14llu ticks  %5.1f %%\n", k == 0 ? "refill + generation" : (k == 1 ? "nearest hit" : "interaction + recording"),
                (unsigned long long)v[28 + k], 100.0 * (double)v[28 + k] / (double)(v[28] + v[29] + v[30]));
    // (grid kernel, ODW_GRID_STATS: the same for its phases)
    uint64_t gtotal = 0;
    for (int k = 24; k < 28; ++k) gtotal += v[k];
    const char* gphases[4] = {"A refill", "B setup", "C cell steps", "D resolve+interact"};
    if (gtotal)
      for (int k = 0; k < 4; ++k)
        fprintf(stderr, "[odw grid time] %-18s %14llu ticks  %5.1f %%\n", gphases[k], (unsigned long long)v[24 + k], 100.0 * (double)v[24 + k] / (double)gtotal);
    release(ctx->dbg);
  }
  for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (auto& ev : ctx->free_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  DevBuf* all[] = {&ctx->prim_f64, &ctx->prim_hdr, &ctx->prim_i32, &ctx->cond_i32, &ctx->group_f64, &ctx->group_i32,
                   &ctx->group_gdir, &ctx->seq_mask, &ctx->bvh_nodes, &ctx->bvh_prims, &ctx->bvh_leaf, &ctx->bvh_wide,
                   &ctx->phi_tab, &ctx->t_tab, &ctx->t_guide, &ctx->d_source, &ctx->d_det, &ctx->hits, &ctx->hit_count, &ctx->chunk_counter, &ctx->results,
                   &ctx->ray_o, &ctx->ray_d, &ctx->ray_p, &ctx->ray_aos, &ctx->samp_t, &ctx->samp_phi,
                   &ctx->sort_keys[0], &ctx->sort_keys[1], &ctx->sort_vals[0], &ctx->sort_vals[1],
                   &ctx->sort_tmp, &ctx->sorted_rows, &ctx->segs, &ctx->seg_count};
  for (DevBuf* b : all) release(*b);
  for (DevBuf* b : {&ctx->em_prim_f64, &ctx->em_prim_i32, &ctx->em_cond, &ctx->em_face_i32, &ctx->em_face_cdf,
                    &ctx->em_t_tab, &ctx->em_t_guide, &ctx->em_o, &ctx->em_d, &ctx->em_tri_nrm, &ctx->tri_nrm, &ctx->phi_guide})
    release(*b);
  for (auto& sb : ctx->surf_bufs) { release(sb.phi_tab); release(sb.t_tab); release(sb.t_guide); }
  release(ctx->d_samplers);
  release(ctx->d_group_sampler);
  for (DevBuf* b : {&ctx->grid_bounds, &ctx->grid_cells, &ctx->grid_items}) release(*b);
  for (DevBuf* b : {&ctx->ph_bitmap, &ctx->ph_before, &ctx->ph_row_of}) release(*b);
  for (DevBuf* b : {&ctx->ph_sel_entering, &ctx->ph_flags, &ctx->ph_x, &ctx->ph_y, &ctx->ph_sorted, &ctx->ph_small,
                    &ctx->ph_part, &ctx->ph_edges, &ctx->ph_edges_b, &ctx->ph_counts, &ctx->ph_sel_hist, &ctx->ph_accel})
    release(*b);
  release(ctx->archive);
  release(ctx->archive_count);
  release(ctx->batch_values);
  release(ctx->batch_rays_buf);
  release(ctx->batch_hits);
  release(ctx->batch_hit_count);
  if (ctx->phb_pin_p) (void)hipHostFree(ctx->phb_pin_p);
  if (ctx->up_pin) (void)hipHostFree(ctx->up_pin);
  if (ctx->phb_ev) (void)hipEventDestroy(ctx->phb_ev);
  for (DevBuf* b : {&ctx->phb_row_of, &ctx->phb_words, &ctx->phb_sel, &ctx->phb_small, &ctx->phb_rows, &ctx->phb_x, &ctx->phb_y,
                    &ctx->phb_part, &ctx->phb_sel_hist, &ctx->phb_cand, &ctx->phb_counts, &ctx->phb_scenes, &ctx->phb_hist, &ctx->phb_planes,
                    &ctx->phb_strides, &ctx->phb_origins, &ctx->phb_accel, &ctx->phb_pts})
    release(*b);
  release(ctx->alt_hits);
  release(ctx->alt_hit_count);
  if (ctx->alt_ready) (void)hipEventDestroy(ctx->alt_ready);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// host half of odw_upload_scene: validation and the host copies of every table (no device call)
static int scene_host_tables(odw_ctx* ctx, const odw_scene_desc* s) {
  if (!ctx || !s) return fail(ctx, ODW_ERR_INVALID, "odw_upload_scene: null argument");
  if (s->n_prims < 0 || s->n_groups < 0 || s->n_groups > ODW_MAX_GROUPS || s->seq_len < 0 ||
      s->seq_len > ODW_MAX_SEQUENCE || s->n_conds < 0 || s->n_conds >= (1 << 24))
    return fail(ctx, ODW_ERR_INVALID, "odw_upload_scene: counts out of range");
  if ((s->n_prims > 0 && (!s->prim_type || !s->prim_group || !s->prim_flags || !s->prim_xform || !s->prim_params ||
                          !s->prim_cond_off)) ||
      (s->n_conds > 0 && (!s->cond_prim || !s->cond_inside)) ||
      (s->n_groups > 0 && (!s->group_type || !s->group_ior || !s->group_refl || !s->group_abslen || !s->group_record)) ||
      (s->seq_len > 0 && !s->seq_mask))
    return fail(ctx, ODW_ERR_INVALID, "odw_upload_scene: null table pointer");
  const int n = s->n_prims;
  int max_solid = 0;
  for (int p = 0; p < n && s->prim_solid; ++p) max_solid = std::max(max_solid, s->prim_solid[p]);
  ctx->h_prim_f64.assign((size_t)n * 16, 0.0);
  ctx->h_prim_i32.assign((size_t)n * 4, 0);
  for (int p = 0; p < n; ++p) {
    const int type = s->prim_type[p], group = s->prim_group[p];
    if (type < ODW_PRIM_BOX || type > ODW_PRIM_PARABOLOID) return fail(ctx, ODW_ERR_UNSUPPORTED, "unknown primitive type");
    if (group < 0 || group >= s->n_groups) return fail(ctx, ODW_ERR_INVALID, "primitive group out of range");
    const int off = s->prim_cond_off[p], cnt = s->prim_cond_off[p + 1] - off;
    if (off < 0 || cnt < 0 || cnt > 255 || off + cnt > s->n_conds)
      return fail(ctx, ODW_ERR_INVALID, "bad condition offsets");
    if (type == ODW_PRIM_TRIANGLE) {
      if (cnt) return fail(ctx, ODW_ERR_UNSUPPORTED, "triangles cannot carry trimming conditions");
      const double* v = s->prim_xform + 12 * (size_t)p;
      double* d = &ctx->h_prim_f64[16 * (size_t)p];
      double e1[3], e2[3], e3[3], nn[3];
      for (int k = 0; k < 3; ++k) { d[k] = v[k]; e1[k] = v[3 + k] - v[k]; e2[k] = v[6 + k] - v[k]; e3[k] = e2[k] - e1[k]; }
      nn[0] = e1[1] * e2[2] - e1[2] * e2[1];
      nn[1] = e1[2] * e2[0] - e1[0] * e2[2];
      nn[2] = e1[0] * e2[1] - e1[1] * e2[0];
      const double a2 = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);   // twice the area
      if (!(a2 > 0) || !std::isfinite(a2)) return fail(ctx, ODW_ERR_INVALID, "degenerate triangle");
      auto len3 = [](const double* x) { return std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); };
      for (int k = 0; k < 3; ++k) { d[3 + k] = e1[k]; d[6 + k] = e2[k]; d[9 + k] = nn[k] / a2; }
      // a point at distance tol outside an edge has barycentric coordinate -tol/altitude
      d[12] = len3(e2) / a2;   // u: distance from edge (v0, v2)
      d[13] = len3(e1) / a2;   // v: distance from edge (v0, v1)
      d[14] = len3(e3) / a2;   // u+v: distance from edge (v1, v2)
      // edges shared with a neighbouring facet of the same face are not widened (sign = marker)
      const int face_edges = s->tri_edges ? s->tri_edges[p] : 7;
      for (int k = 0; k < 3; ++k)
        if (!((face_edges >> k) & 1)) d[12 + k] = -d[12 + k];
      d[15] = 0.0;
    } else {
      std::memcpy(&ctx->h_prim_f64[16 * (size_t)p], s->prim_xform + 12 * (size_t)p, 12 * sizeof(double));
      std::memcpy(&ctx->h_prim_f64[16 * (size_t)p + 12], s->prim_params + 4 * (size_t)p, 4 * sizeof(double));
      if (type == ODW_PRIM_SPHERE) {
        // the kernel intersects spheres without their frame: centre in global coordinates = -R^T t
        const double* m = &ctx->h_prim_f64[16 * (size_t)p];
        for (int k = 0; k < 3; ++k)
          ctx->h_prim_f64[16 * (size_t)p + 13 + k] = -(m[k] * m[3] + m[4 + k] * m[7] + m[8 + k] * m[11]);
      }
      if (type == ODW_PRIM_PARABOLOID) {
        double* par = &ctx->h_prim_f64[16 * (size_t)p + 12];
        if (!(par[0] > 0) || !(par[1] > 0)) return fail(ctx, ODW_ERR_INVALID, "paraboloid: focal length and height must be positive");
        par[2] = 2.0 * std::sqrt(par[0] * par[1]);            // rim radius at z = H
      }
    }
    ctx->h_prim_i32[4 * p] = type;
    ctx->h_prim_i32[4 * p + 1] = group;
    // flags | facemask << 8 in the low half, solid id above; scenes with more solids than fit lose the
    // convex-solid shortcut, nothing else
    const int solid = s->prim_solid ? s->prim_solid[p] : 0;
    const bool fits = s->prim_solid && solid >= 0 && solid < 0x7fff && max_solid < 0x7fff;
    ctx->h_prim_i32[4 * p + 2] = ((s->prim_flags[p] & 0xffff & ~ODW_FLAG_ISOLATED) & (fits ? ~0 : ~ODW_FLAG_CONVEX)) | ((fits ? solid : 0x7fff) << ODW_SOLID_SHIFT);
    ctx->h_prim_i32[4 * p + 3] = off | (cnt << 24);
  }
  std::vector<int32_t> cond((size_t)std::max(1, s->n_conds), 0);
  for (int c = 0; c < s->n_conds; ++c) {
    if (s->cond_prim[c] < 0 || s->cond_prim[c] >= n) return fail(ctx, ODW_ERR_INVALID, "condition primitive out of range");
    if (s->prim_type[s->cond_prim[c]] == ODW_PRIM_TRIANGLE)
      return fail(ctx, ODW_ERR_UNSUPPORTED, "trimming against a triangle (no inside/outside of a facet)");
    cond[c] = s->cond_prim[c] | (s->cond_inside[c] ? (int32_t)0x80000000 : 0);
  }
  ctx->h_cond = cond;
  std::vector<double> gf(ODW_MAX_GROUPS * 4, 0.0), gd(ODW_MAX_GROUPS * 3, 0.0);
  std::vector<int32_t> gi(ODW_MAX_GROUPS * 4, 0);
  ctx->lean = getenv("ODW_NO_LEAN") == nullptr;
  for (int g = 0; g < s->n_groups; ++g)
    if (s->group_type[g] == ODW_OPT_GRATING || !(s->group_abslen[g] == INFINITY)) ctx->lean = false;
  for (int g = 0; g < s->n_groups; ++g) {
    if (s->group_type[g] < ODW_OPT_MIRROR || s->group_type[g] > ODW_OPT_VACUUM)
      return fail(ctx, ODW_ERR_INVALID, "unknown optical type");
    gf[4 * g] = s->group_ior[g];
    gf[4 * g + 1] = s->group_refl[g];
    gf[4 * g + 2] = s->group_abslen[g];
    gf[4 * g + 3] = s->group_grating_lpm ? s->group_grating_lpm[g] : 1000.0;
    gi[4 * g] = s->group_type[g];
    gi[4 * g + 1] = s->group_record[g] ? 1 : 0;
    gi[4 * g + 2] = s->group_grating_type ? s->group_grating_type[g] : 0;
    gi[4 * g + 3] = s->group_grating_order ? s->group_grating_order[g] : 1;
    for (int k = 0; k < 3; ++k) gd[3 * g + k] = s->group_grating_dir ? s->group_grating_dir[3 * g + k] : (k == 2);
  }
  std::vector<uint64_t> seq((size_t)std::max(1, s->seq_len), 0);
  for (int i = 0; i < s->seq_len; ++i) seq[i] = s->seq_mask[i];
  ctx->h_group_f64 = gf;
  ctx->h_group_i32 = gi;
  ctx->h_group_gdir = gd;
  ctx->h_seq = seq;
  DeviceScene& d = ctx->P.scene;
  d.n_prims = n;
  d.n_groups = s->n_groups;
  d.n_nodes = 0;
  d.seq_enabled = s->seq_enabled ? 1 : 0;
  d.seq_len = s->seq_len;
  d.all_mask = (s->n_groups >= 64) ? ~0ull : ((1ull << s->n_groups) - 1ull);
  d.ignore_mask = s->ignore_mask;
  return ODW_OK;
}

int odw_upload_scene(odw_ctx* ctx, const odw_scene_desc* s) {
  int rc = scene_host_tables(ctx, s);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int n = s->n_prims;
  const std::vector<int32_t>& cond = ctx->h_cond;
  const std::vector<double>&gf = ctx->h_group_f64, &gd = ctx->h_group_gdir;
  const std::vector<int32_t>& gi = ctx->h_group_i32;
  const std::vector<uint64_t>& seq = ctx->h_seq;
  if ((rc = upload(ctx, ctx->prim_f64, ctx->h_prim_f64.data(), ctx->h_prim_f64.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->prim_i32, ctx->h_prim_i32.data(), ctx->h_prim_i32.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload(ctx, ctx->cond_i32, cond.data(), cond.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload(ctx, ctx->group_f64, gf.data(), gf.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->group_i32, gi.data(), gi.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload(ctx, ctx->group_gdir, gd.data(), gd.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->seq_mask, seq.data(), seq.size() * sizeof(uint64_t)))) return rc;
  if (s->tri_normals && n > 0) {
    if ((rc = upload(ctx, ctx->tri_nrm, s->tri_normals, (size_t)n * 9 * sizeof(double)))) return rc;
  }
  if ((rc = upload_done(ctx))) return rc;
  DeviceScene& d = ctx->P.scene;
  d.tri_nrm = (s->tri_normals && n > 0) ? (const double*)ctx->tri_nrm.p : nullptr;
  d.prim_f64 = (const double*)ctx->prim_f64.p;
  d.prim_i32 = (const int32_t*)ctx->prim_i32.p;
  d.cond_i32 = (const int32_t*)ctx->cond_i32.p;
  d.group_f64 = (const double*)ctx->group_f64.p;
  d.group_i32 = (const int32_t*)ctx->group_i32.p;
  d.group_gdir = (const double*)ctx->group_gdir.p;
  d.seq_mask = (const uint64_t*)ctx->seq_mask.p;
  ctx->have_scene = true;
  ctx->bvh_dirty = true;
  ctx->spec_dirty = true;
  ctx->spec_fn = nullptr;
  ctx->n_samplers = 0;   // surface samplers belong to the previous scene's groups
  ctx->batch_n = 0;      // an uploaded batch belonged to the previous scene (odw_upload_scene_batch sets it again after this call)
  return ODW_OK;
}

int odw_compile_scene(odw_ctx* ctx, int32_t mode) {
  if (!ctx || mode < ODW_COMPILE_OFF || mode > ODW_COMPILE_AUTO) return fail(ctx, ODW_ERR_INVALID, "odw_compile_scene: bad argument");
  ctx->compile_mode = mode;
  ctx->spec_dirty = true;
  ctx->spec_fn = nullptr;
  if (mode == ODW_COMPILE_OFF || !ctx->have_scene || !ctx->have_limits) return ODW_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->bvh_dirty) {
    int rc = build_bvh(ctx);
    if (rc) return rc;
  }
  return spec_bind(ctx);
}

int odw_compiled_info(odw_ctx* ctx, int32_t* bound, double* compile_seconds, int32_t* cache_hit) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_compiled_info: null context");
  if (bound) *bound = (ctx->spec_fn && !ctx->spec_dirty && !ctx->bvh_dirty) ? ctx->compile_mode : 0;
  if (compile_seconds) *compile_seconds = ctx->spec_seconds;
  if (cache_hit) *cache_hit = ctx->spec_cache_hit;
  return ODW_OK;
}

int odw_compile_check(const odw_scene_desc* scene, const odw_limits* limits, int32_t mode, const char* arch,
                      char* header_out, uint64_t header_capacity, uint64_t* code_bytes) {
  if (!scene || !limits || mode != ODW_COMPILE_STRUCTURE)
    return fail(nullptr, ODW_ERR_INVALID, "odw_compile_check: bad argument");
  odw_ctx tmp;                     // host tables only: no device, no stream
  std::memset(&tmp.P, 0, sizeof tmp.P);
  int rc = scene_host_tables(&tmp, scene);
  if (rc) return rc;
  tmp.P.lim.dist_tol = limits->dist_tol;
  tmp.have_limits = true;
  std::vector<Box> boxes;
  std::vector<char> dead;
  compute_boxes(&tmp, boxes, dead);
  const std::string why = spec_ineligible(&tmp);
  if (!why.empty()) return fail(nullptr, ODW_ERR_UNSUPPORTED, "odw_compile_check: " + why);
  const std::string text = spec_text(&tmp);
  if (header_out && header_capacity) {
    const size_t k = std::min<size_t>(text.size(), (size_t)header_capacity - 1);
    std::memcpy(header_out, text.data(), k);
    header_out[k] = 0;
  }
  std::vector<char> code;
  std::string err;
  if (!spec_compile(text, arch && *arch ? arch : "gfx950", code, err))
    return fail(nullptr, ODW_ERR_DEVICE, err);
  if (code_bytes) *code_bytes = code.size();
  return ODW_OK;
}

// The host half of odw_upload_scene + the first launch's preparations -- validation, host tables, bounding boxes, the
// choice among the flat loop, the rectilinear grid, the binary tree and the eight-wide tree with its leaf records, and
// their construction -- on a context WITHOUT a device: every table the builders would upload goes to host memory instead.
// For tests of these 1 500 lines under a CPU sanitizer (tests/test_native_sanitized.py) and for callers that want to know
// what a scene will be traced with before a GPU is there.  structure: 0 flat loop, 1 grid, 2 binary tree, 3 eight-wide
// tree (facets); sizes: [primitives, tree nodes, grid cells, grid items, bytes of dynamic LDS of a grid block, dead primitives].
int odw_build_check(const odw_scene_desc* scene, const odw_limits* limits, int32_t* structure, uint64_t* sizes) {
  if (!scene || !limits) return fail(nullptr, ODW_ERR_INVALID, "odw_build_check: null argument");
  if (!(limits->dist_tol > 0) || limits->max_intersections < 0 || !(limits->max_ray_length > 0))
    return fail(nullptr, ODW_ERR_INVALID, "odw_build_check: limits out of range");
  odw_ctx tmp;
  std::memset(&tmp.P, 0, sizeof tmp.P);
  tmp.host_only = true;
  if (const char* e = getenv("ODW_BVH_THRESHOLD")) tmp.flat_limit = atoi(e);
  int rc = scene_host_tables(&tmp, scene);
  if (!rc) {
    tmp.P.lim.max_ray_length = limits->max_ray_length;
    tmp.P.lim.max_intersections = limits->max_intersections;
    tmp.P.lim.dist_tol = limits->dist_tol;
    tmp.P.lim.power_tol = limits->power_tol;
    tmp.have_limits = tmp.have_scene = true;
    if (scene->tri_normals && scene->n_prims > 0)
      rc = upload(&tmp, tmp.tri_nrm, scene->tri_normals, (size_t)scene->n_prims * 9 * sizeof(double));
    if (!rc) rc = build_bvh(&tmp);
  }
  if (!rc) {
    if (structure) *structure = tmp.P.grid.nx > 0 ? 1 : (tmp.P.scene.n_nodes ? (tmp.P.scene.bvh_leaf ? 3 : 2) : 0);
    if (sizes) {
      uint64_t dead = 0;
      for (char d : tmp.h_dead) dead += d ? 1 : 0;
      sizes[0] = (uint64_t)tmp.P.scene.n_prims;
      sizes[1] = (uint64_t)tmp.P.scene.n_nodes;
      sizes[2] = (uint64_t)tmp.P.grid.nx * (uint64_t)tmp.P.grid.ny * (uint64_t)tmp.P.grid.nz;
      sizes[3] = (uint64_t)tmp.P.grid.n_items;
      sizes[4] = (uint64_t)tmp.P.grid.lds_bytes;
      sizes[5] = dead;
    }
  } else {
    g_error = tmp.err;
  }
  for (DevBuf* b : {&tmp.prim_f64, &tmp.prim_hdr, &tmp.prim_i32, &tmp.cond_i32, &tmp.group_f64, &tmp.group_i32, &tmp.group_gdir, &tmp.seq_mask,
                    &tmp.bvh_nodes, &tmp.bvh_prims, &tmp.bvh_leaf, &tmp.bvh_wide, &tmp.tri_nrm, &tmp.grid_bounds, &tmp.grid_cells, &tmp.grid_items})
    release(*b);
  return rc;
}

int odw_upload_surface_samplers(odw_ctx* ctx, const odw_surface_sampler_desc* samplers, int32_t n) {
  if (!ctx || n < 0 || (n && !samplers)) return fail(ctx, ODW_ERR_INVALID, "odw_upload_surface_samplers: bad argument");
  if (!ctx->have_scene) return fail(ctx, ODW_ERR_NO_SCENE, "odw_upload_surface_samplers before odw_upload_scene");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n == 0 && ctx->n_samplers == 0) return ODW_OK;   // (none before, none now: nothing to wait for, nothing to bind again)
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // a running launch may still read the old tables
  ctx->n_samplers = 0;
  ctx->spec_dirty = true;                           // (a compiled scene: the kernel variant with / without scatter())
  if (n == 0) return ODW_OK;
  std::vector<int32_t> gs(ODW_MAX_GROUPS * 2, -1);
  std::vector<DeviceSurfaceSampler> ds((size_t)n);
  if ((int)ctx->surf_bufs.size() < n) ctx->surf_bufs.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    const odw_surface_sampler_desc& s = samplers[i];
    if (s.group < 0 || s.group >= ctx->P.scene.n_groups || s.kind < ODW_SURF_PRIMARY || s.kind > ODW_SURF_MODIFY)
      return fail(ctx, ODW_ERR_INVALID, "surface sampler: group/kind out of range");
    // several samplers of one (group, kind): told apart by mu (a chain: group_sampler -> first, DeviceSurfaceSampler::next)
    for (int j = gs[2 * s.group + s.kind]; j >= 0; j = ds[(size_t)j].next)
      if (ds[(size_t)j].mu == s.mu) return fail(ctx, ODW_ERR_INVALID, "surface sampler: duplicate (group, kind, mu)");
    if (s.n_atoms < 0 || s.n_atoms > ODW_SURF_MAX_ATOMS || (s.n_atoms && (!s.atom_mass || !s.atom_theta || !s.atom_phi)))
      return fail(ctx, ODW_ERR_INVALID, "surface sampler: atoms");
    for (int k = 0; k < s.n_family * s.n_atoms; ++k)
      if (!(s.atom_mass[k] >= 0.0 && s.atom_mass[k] <= 1.0)) return fail(ctx, ODW_ERR_INVALID, "surface sampler: atom probability outside [0, 1]");
    if (s.n_family < 1 || s.family_axis < ODW_SURF_AXIS_NONE || s.family_axis > ODW_SURF_AXIS_THETA_REFL ||
        (s.family_axis == ODW_SURF_AXIS_NONE && s.n_family != 1) ||
        (s.n_family > 1 && !(s.family_hi > s.family_lo)))
      return fail(ctx, ODW_ERR_INVALID, "surface sampler: family");
    if (s.n_phi_knots < 2 || s.n_t_knots < 2 || s.n_t_rows < 1 ||
        (s.n_t_rows != 1 && s.n_t_rows != s.n_phi_knots - 1))
      return fail(ctx, ODW_ERR_INVALID, "surface sampler: table shape");
    if (!s.phi_edges || !s.phi_cdf || !s.t_edges || !s.t_cdf)
      return fail(ctx, ODW_ERR_INVALID, "surface sampler: null table pointer");
    const size_t np = (size_t)s.n_phi_knots, nt = (size_t)s.n_t_knots, rows = (size_t)s.n_t_rows, nf = (size_t)s.n_family;
    std::vector<double> ptab(nf * np * 2), ttab(nf * rows * nt * 2);
    std::vector<int32_t> guide(nf * rows * (kSurfaceGuide + 1));
    for (size_t k = 0; k < nf; ++k) {
      const double* pc = s.phi_cdf + k * np;
      if (pc[0] != 0.0 || pc[np - 1] != 1.0) return fail(ctx, ODW_ERR_INVALID, "surface sampler: phi cdf must run from 0 to 1");
      for (size_t j = 0; j < np; ++j) {
        if (j && pc[j] < pc[j - 1]) return fail(ctx, ODW_ERR_INVALID, "surface sampler: cdf not monotone");
        ptab[(k * np + j) * 2] = pc[j];
        ptab[(k * np + j) * 2 + 1] = s.phi_edges[j];
      }
      for (size_t r = 0; r < rows; ++r) {
        const double* cdf = s.t_cdf + (k * rows + r) * nt;
        if (cdf[0] != 0.0 || cdf[nt - 1] != 1.0) return fail(ctx, ODW_ERR_INVALID, "surface sampler: theta cdf rows must run from 0 to 1");
        double* dst = ttab.data() + (k * rows + r) * nt * 2;
        for (size_t j = 0; j < nt; ++j) {
          if (j && cdf[j] < cdf[j - 1]) return fail(ctx, ODW_ERR_INVALID, "surface sampler: cdf not monotone");
          dst[2 * j] = cdf[j];
          dst[2 * j + 1] = s.t_edges[j];
        }
        int32_t* g = guide.data() + (k * rows + r) * (kSurfaceGuide + 1);
        size_t j = 0;
        for (int q = 0; q <= kSurfaceGuide; ++q) {
          const double x = (double)q / (double)kSurfaceGuide;
          while (j + 1 < nt && cdf[j + 1] <= x) ++j;
          g[q] = (int32_t)j;
        }
      }
    }
    odw_ctx::SurfaceBufs& sb = ctx->surf_bufs[(size_t)i];
    int rc;
    if ((rc = upload(ctx, sb.phi_tab, ptab.data(), ptab.size() * sizeof(double)))) return rc;
    if ((rc = upload(ctx, sb.t_tab, ttab.data(), ttab.size() * sizeof(double)))) return rc;
    if ((rc = upload(ctx, sb.t_guide, guide.data(), guide.size() * sizeof(int32_t)))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // staging vectors go out of scope
    DeviceSurfaceSampler& d = ds[(size_t)i];
    d.phi_tab = (const double*)sb.phi_tab.p;
    d.t_tab = (const double*)sb.t_tab.p;
    d.t_guide = (const int32_t*)sb.t_guide.p;
    d.n_phi_knots = s.n_phi_knots;
    d.n_t_knots = s.n_t_knots;
    d.n_t_rows = s.n_t_rows;
    d.n_guide = kSurfaceGuide;
    d.axis = s.family_axis;
    d.n_family = s.n_family;
    d.lo = s.family_lo;
    d.inv_step = s.n_family > 1 ? (double)(s.n_family - 1) / (s.family_hi - s.family_lo) : 0.0;
    d.mu = s.mu;
    d.n_atoms = s.n_atoms;
    d.atom_mass = nullptr;
    std::memset(d.atom_theta, 0, sizeof d.atom_theta);
    std::memset(d.atom_phi, 0, sizeof d.atom_phi);
    if (s.n_atoms) {
      if ((rc = upload(ctx, sb.atom_mass, s.atom_mass, (size_t)s.n_family * (size_t)s.n_atoms * sizeof(double)))) return rc;
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      d.atom_mass = (const double*)sb.atom_mass.p;
      for (int j = 0; j < s.n_atoms; ++j)
        for (int c = 0; c < 3; ++c) { d.atom_theta[j][c] = s.atom_theta[3 * j + c]; d.atom_phi[j][c] = s.atom_phi[3 * j + c]; }
    }
    d.next = gs[2 * s.group + s.kind];       // (the chain runs from the last one uploaded to the first)
    gs[2 * s.group + s.kind] = i;
  }
  int rc;
  if ((rc = upload(ctx, ctx->d_samplers, ds.data(), ds.size() * sizeof(DeviceSurfaceSampler)))) return rc;
  if ((rc = upload(ctx, ctx->d_group_sampler, gs.data(), gs.size() * sizeof(int32_t)))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->n_samplers = n;
  return ODW_OK;
}

int odw_set_wavelength(odw_ctx* ctx, double wavelength_nm) {
  if (!ctx || !(wavelength_nm > 0)) return fail(ctx, ODW_ERR_INVALID, "odw_set_wavelength: bad argument");
  ctx->P.wavelength = wavelength_nm;
  return ODW_OK;
}

int odw_set_surface_seed(odw_ctx* ctx, uint64_t seed) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_set_surface_seed: null ctx");
  ctx->surface_seed = seed;
  return ODW_OK;
}

int odw_upload_source(odw_ctx* ctx, const odw_source_desc* s) {
  if (!ctx || !s) return fail(ctx, ODW_ERR_INVALID, "odw_upload_source: null argument");
  if (s->n_phi_knots < 2 || s->n_t_knots < 2 || s->n_t_rows < 1 ||
      (s->n_t_rows != 1 && s->n_t_rows != s->n_phi_knots - 1))
    return fail(ctx, ODW_ERR_INVALID, "odw_upload_source: table shape");
  if (!s->phi_edges || !s->phi_cdf || !s->t_edges || !s->t_cdf)
    return fail(ctx, ODW_ERR_INVALID, "odw_upload_source: null table pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int np = s->n_phi_knots, nt = s->n_t_knots, rows = s->n_t_rows;
  if (s->phi_cdf[0] != 0.0 || s->phi_cdf[np - 1] != 1.0)
    return fail(ctx, ODW_ERR_INVALID, "phi cdf must run from 0 to 1");
  std::vector<double> ptab((size_t)np * 2), ttab((size_t)rows * nt * 2);
  for (int i = 0; i < np; ++i) { ptab[2 * i] = s->phi_cdf[i]; ptab[2 * i + 1] = s->phi_edges[i]; }
  std::vector<int32_t> guide((size_t)rows * (kGuide + 1));
  for (int r = 0; r < rows; ++r) {
    const double* cdf = s->t_cdf + (size_t)r * nt;
    if (cdf[0] != 0.0 || cdf[nt - 1] != 1.0) return fail(ctx, ODW_ERR_INVALID, "theta cdf rows must run from 0 to 1");
    double* dst = ttab.data() + (size_t)r * nt * 2;
    for (int i = 0; i < nt; ++i) {
      if (i && cdf[i] < cdf[i - 1]) return fail(ctx, ODW_ERR_INVALID, "cdf not monotone");
      dst[2 * i] = cdf[i];
      dst[2 * i + 1] = s->t_edges[i];
    }
    // guide[k] = last knot with cdf <= k/G: brackets the search for any u in
    // [k/G, (k+1)/G) without changing which knot is found
    int32_t* g = guide.data() + (size_t)r * (kGuide + 1);
    int j = 0;
    for (int k = 0; k <= kGuide; ++k) {
      const double x = (double)k / (double)kGuide;
      while (j + 1 < nt && cdf[j + 1] <= x) ++j;
      g[k] = j;
    }
  }
  int rc;
  if ((rc = upload(ctx, ctx->phi_tab, ptab.data(), ptab.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->t_tab, ttab.data(), ttab.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->t_guide, guide.data(), guide.size() * sizeof(int32_t)))) return rc;
  std::vector<int32_t> pguide(kPhiGuide + 1);
  for (int k = 0, j = 0; k <= kPhiGuide; ++k) {
    const double x = (double)k / (double)kPhiGuide;
    while (j + 1 < np && s->phi_cdf[j + 1] <= x) ++j;
    pguide[k] = j;
  }
  if ((rc = upload(ctx, ctx->phi_guide, pguide.data(), pguide.size() * sizeof(int32_t)))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  DeviceSource& d = ctx->h_source;
  std::memcpy(d.m, s->xform, sizeof d.m);
  d.focal_length = s->focal_length;
  d.finite_focal = std::isfinite(s->focal_length) ? 1 : 0;
  d.wavelength = s->wavelength;
  d.power = s->power;
  d.phi_tab = (const double*)ctx->phi_tab.p;
  d.t_tab = (const double*)ctx->t_tab.p;
  d.t_guide = (const int32_t*)ctx->t_guide.p;
  d.phi_guide = (const int32_t*)ctx->phi_guide.p;
  d.n_phi_guide = kPhiGuide;
  d.n_phi_knots = np;
  d.n_t_knots = nt;
  d.n_t_rows = rows;
  d.n_guide = kGuide;
  if ((rc = upload(ctx, ctx->d_source, &ctx->h_source, sizeof(DeviceSource)))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->P.source = (const DeviceSource*)ctx->d_source.p;
  ctx->P.wavelength = s->wavelength;
  ctx->have_source = true;
  ctx->emitter_active = false;
  return ODW_OK;
}

int odw_upload_surface_source(odw_ctx* ctx, const odw_surface_source_desc* s) {
  if (!ctx || !s) return fail(ctx, ODW_ERR_INVALID, "odw_upload_surface_source: null argument");
  if (s->n_prims < 1 || s->n_faces < 1 || s->n_conds < 0 || s->n_t_knots < 2 || !(s->dist_tol > 0))
    return fail(ctx, ODW_ERR_INVALID, "odw_upload_surface_source: counts out of range");
  if (!s->prim_type || !s->prim_flags || !s->prim_xform || !s->prim_params || !s->prim_cond_off || !s->face_prim ||
      !s->face_id || !s->face_area || !s->t_edges || !s->t_cdf || (s->n_conds > 0 && (!s->cond_prim || !s->cond_inside)))
    return fail(ctx, ODW_ERR_INVALID, "odw_upload_surface_source: null table pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const int n = s->n_prims;
  std::vector<double> pf((size_t)n * 16);
  std::vector<int32_t> pi((size_t)n * 4);
  for (int p = 0; p < n; ++p) {
    if (s->prim_type[p] < ODW_PRIM_BOX || s->prim_type[p] > ODW_PRIM_PARABOLOID)
      return fail(ctx, ODW_ERR_UNSUPPORTED, "surface source: unknown primitive kind");
    const int off = s->prim_cond_off[p], cnt = s->prim_cond_off[p + 1] - off;
    if (s->prim_type[p] == ODW_PRIM_TRIANGLE && cnt != 0)
      return fail(ctx, ODW_ERR_INVALID, "surface source: facets cannot carry trimming conditions");
    if (off < 0 || cnt < 0 || off + cnt > s->n_conds) return fail(ctx, ODW_ERR_INVALID, "surface source: bad condition offsets");
    std::memcpy(&pf[16 * (size_t)p], s->prim_xform + 12 * (size_t)p, 12 * sizeof(double));
    std::memcpy(&pf[16 * (size_t)p + 12], s->prim_params + 4 * (size_t)p, 4 * sizeof(double));
    pi[4 * p] = s->prim_type[p];
    pi[4 * p + 1] = s->prim_flags[p];
    pi[4 * p + 2] = off;
    pi[4 * p + 3] = cnt;
  }
  std::vector<int32_t> cond((size_t)std::max(1, s->n_conds), 0);
  for (int c = 0; c < s->n_conds; ++c) {
    if (s->cond_prim[c] < 0 || s->cond_prim[c] >= n || s->prim_type[s->cond_prim[c]] == ODW_PRIM_TRIANGLE)
      return fail(ctx, ODW_ERR_INVALID, "surface source: condition primitive out of range");
    cond[c] = s->cond_prim[c] | (s->cond_inside[c] ? (int32_t)0x80000000 : 0);
  }
  static const int n_faces_of[7] = {6, 1, 3, 3, 1, 1, 0};     // (paraboloid faces do not emit: rejected below)
  std::vector<int32_t> fi((size_t)s->n_faces * 2);
  std::vector<double> fc((size_t)s->n_faces + 1, 0.0);
  double total = 0;
  for (int f = 0; f < s->n_faces; ++f) {
    const int p = s->face_prim[f];
    if (p < 0 || p >= n || s->face_id[f] < 0 || s->face_id[f] >= n_faces_of[s->prim_type[p]] || !(s->face_area[f] >= 0))
      return fail(ctx, ODW_ERR_INVALID, "surface source: face out of range");
    fi[2 * f] = p;
    fi[2 * f + 1] = s->face_id[f];
    total += s->face_area[f];
  }
  if (!(total > 0)) return fail(ctx, ODW_ERR_INVALID, "surface source: emitting faces have no area");
  double run = 0;
  for (int f = 0; f < s->n_faces; ++f) { run += s->face_area[f]; fc[f + 1] = run / total; }
  fc[s->n_faces] = 1.0;
  const int nt = s->n_t_knots;
  if (s->t_cdf[0] != 0.0 || s->t_cdf[nt - 1] != 1.0) return fail(ctx, ODW_ERR_INVALID, "surface source: theta cdf must run from 0 to 1");
  std::vector<double> ttab((size_t)nt * 2);
  for (int i = 0; i < nt; ++i) {
    if (i && s->t_cdf[i] < s->t_cdf[i - 1]) return fail(ctx, ODW_ERR_INVALID, "surface source: cdf not monotone");
    ttab[2 * i] = s->t_cdf[i];
    ttab[2 * i + 1] = s->t_edges[i];
  }
  std::vector<int32_t> guide((size_t)kGuide + 1);
  for (int k = 0, j = 0; k <= kGuide; ++k) {
    const double x = (double)k / (double)kGuide;
    while (j + 1 < nt && s->t_cdf[j + 1] <= x) ++j;
    guide[k] = j;
  }
  int rc;
  if ((rc = upload(ctx, ctx->em_prim_f64, pf.data(), pf.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->em_prim_i32, pi.data(), pi.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload(ctx, ctx->em_cond, cond.data(), cond.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload(ctx, ctx->em_face_i32, fi.data(), fi.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload(ctx, ctx->em_face_cdf, fc.data(), fc.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->em_t_tab, ttab.data(), ttab.size() * sizeof(double)))) return rc;
  if ((rc = upload(ctx, ctx->em_t_guide, guide.data(), guide.size() * sizeof(int32_t)))) return rc;
  if (s->tri_normals && (rc = upload(ctx, ctx->em_tri_nrm, s->tri_normals, (size_t)n * 9 * sizeof(double)))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  DeviceEmitter& e = ctx->h_emitter;
  e.tri_nrm = s->tri_normals ? (const double*)ctx->em_tri_nrm.p : nullptr;
  e.prim_f64 = (const double*)ctx->em_prim_f64.p;
  e.prim_i32 = (const int32_t*)ctx->em_prim_i32.p;
  e.cond_i32 = (const int32_t*)ctx->em_cond.p;
  e.face_i32 = (const int32_t*)ctx->em_face_i32.p;
  e.face_cdf = (const double*)ctx->em_face_cdf.p;
  e.t_tab = (const double*)ctx->em_t_tab.p;
  e.t_guide = (const int32_t*)ctx->em_t_guide.p;
  e.n_faces = s->n_faces;
  e.n_t_knots = nt;
  e.n_guide = kGuide;
  e.dist_tol = s->dist_tol;
  e.wavelength = s->wavelength;
  e.power = s->power;
  ctx->P.wavelength = s->wavelength;
  ctx->have_source = true;
  ctx->emitter_active = true;
  return ODW_OK;
}

int odw_generate_rays(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, uint64_t seed, double* origins,
                      double* directions) {
  if (!ctx || !origins || !directions) return fail(ctx, ODW_ERR_INVALID, "odw_generate_rays: bad argument");
  if (!ctx->have_source) return fail(ctx, ODW_ERR_NO_SCENE, "source not uploaded");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  for (uint64_t off = 0; off < n_rays; off += kEmitChunk) {
    const uint64_t m = std::min<uint64_t>(kEmitChunk, n_rays - off);
    int rc;
    if (ctx->emitter_active) {
      if ((rc = emit_rays(ctx, first_ray + off, m, seed, false))) return rc;
    } else {
      if ((rc = ensure(ctx, ctx->em_o, m * 3 * sizeof(double)))) return rc;
      if ((rc = ensure(ctx, ctx->em_d, m * 3 * sizeof(double)))) return rc;
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, (uint64_t)ctx->n_cu * 16));
      hipLaunchKernelGGL(odw_make_rays_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->P.source, first_ray + off,
                         m, seed, (double*)ctx->em_o.p, (double*)ctx->em_d.p);
      HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(origins + 3 * off, ctx->em_o.p, m * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(directions + 3 * off, ctx->em_d.p, m * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return ODW_OK;
}

int odw_set_limits(odw_ctx* ctx, const odw_limits* l) {
  if (!ctx || !l) return fail(ctx, ODW_ERR_INVALID, "odw_set_limits: null argument");
  if (!(l->dist_tol > 0) || l->max_intersections < 0 || !(l->max_ray_length > 0))
    return fail(ctx, ODW_ERR_INVALID, "odw_set_limits: values out of range");
  if (!ctx->have_limits || ctx->P.lim.dist_tol != l->dist_tol) {
    ctx->bvh_dirty = true;   // (the boxes carry the tolerance; a compiled scene is bound again after the rebuild)
    ctx->batch_n = 0;        // (and so do the boxes of an uploaded batch: it has to be uploaded again)
  }
  ctx->P.lim.max_ray_length = l->max_ray_length;
  ctx->P.lim.max_intersections = l->max_intersections;
  ctx->P.lim.dist_tol = l->dist_tol;
  ctx->P.lim.power_tol = l->power_tol;
  ctx->have_limits = true;
  return ODW_OK;
}

int odw_set_detector(odw_ctx* ctx, const odw_detector_desc* det) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_set_detector: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DeviceDetector& d = ctx->h_det;
  if (!det) { d.enabled = 0; ctx->P.det_enabled = 0; ctx->n_bins = 0; return ODW_OK; }
  if (det->nx <= 0 || det->ny <= 0 || !(det->x_hi > det->x_lo) || !(det->y_hi > det->y_lo))
    return fail(ctx, ODW_ERR_INVALID, "odw_set_detector: bad window");
  ctx->det_desc = *det;
  for (int k = 0; k < 3; ++k) { d.origin[k] = det->origin[k]; d.ex[k] = det->ex[k]; d.ey[k] = det->ey[k]; }
  d.x_lo = det->x_lo;
  d.y_lo = det->y_lo;
  d.x_scale = det->nx / (det->x_hi - det->x_lo);
  d.y_scale = det->ny / (det->y_hi - det->y_lo);
  d.nx = det->nx;
  d.ny = det->ny;
  d.nx_f = (double)det->nx;
  d.ny_f = (double)det->ny;
  d.group = det->group;
  d.enabled = 1;
  ctx->n_bins = (uint64_t)det->nx * (uint64_t)det->ny;
  int rc = ensure_results(ctx, ctx->n_bins);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // a running launch may still read the old block
  if ((rc = upload(ctx, ctx->d_det, &ctx->h_det, sizeof(DeviceDetector)))) return rc;
  ctx->P.det = (const DeviceDetector*)ctx->d_det.p;
  ctx->P.det_enabled = 1;
  HIPCHK(ctx, hipMemsetAsync(ctx->hist.p, 0, ctx->n_bins * sizeof(uint64_t), ctx->stream));
  return ODW_OK;
}

int odw_reserve_hits(odw_ctx* ctx, uint64_t capacity) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_reserve_hits: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (capacity == 0) { release(ctx->hits); ctx->hit_capacity = ctx->hit_slots = 0; return ODW_OK; }
  if (capacity > ctx->hit_capacity) {
    release(ctx->hits);
    ctx->hit_capacity = ctx->hit_slots = 0;
    // big lists get slack for block reservations (launch_trace): an eighth (unused slots at block
    // changes) + one block per wave of the largest grid
    uint64_t slots = capacity;
    // (a short list is filled by short launches: a wave per 256 rows, the largest grid at most)
    if (capacity >= kHitBlockMinRows)
      slots += hit_block_room(capacity, std::min<uint64_t>((uint64_t)ctx->n_cu * 8 * 4, std::max<uint64_t>(64, capacity / 256)), kHitBlock);
    int rc = ensure(ctx, ctx->hits, slots * sizeof(odw_hit));
    if (rc) return rc;
    ctx->hit_capacity = capacity;
    ctx->hit_slots = slots;
    // a new buffer: rows recorded so far are gone
    HIPCHK(ctx, hipMemsetAsync(ctx->hit_count.p, 0, 2 * sizeof(uint64_t), ctx->stream));
  }
  return ODW_OK;
}

int odw_reserve_segments(odw_ctx* ctx, uint64_t capacity) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_reserve_segments: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (capacity == 0) { release(ctx->segs); ctx->seg_capacity = 0; return ODW_OK; }
  if (capacity > 0x7FFFFFFFull) return fail(ctx, ODW_ERR_CAPACITY, "odw_reserve_segments: more than 2^31 rows");
  if (capacity > ctx->seg_capacity) {
    release(ctx->segs);
    ctx->seg_capacity = 0;
    int rc = ensure(ctx, ctx->segs, capacity * sizeof(odw_segment));
    if (rc) return rc;
    ctx->seg_capacity = capacity;
    HIPCHK(ctx, hipMemsetAsync(ctx->seg_count.p, 0, sizeof(uint64_t), ctx->stream));
  }
  return ODW_OK;
}

int odw_trace(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, uint64_t seed, uint32_t flags) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_trace: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  if (!ctx->emitter_active) return launch_trace(ctx, first_ray, n_rays, seed, flags, nullptr, nullptr, nullptr);
  // surface source: initial conditions are generated into a staging buffer,
  // kEmitChunk rays at a time, and traced as explicit rays (same stream: the
  // next chunk's generation waits for the previous chunk's trace)
  for (uint64_t off = 0; off < n_rays; off += kEmitChunk) {
    const uint64_t m = std::min<uint64_t>(kEmitChunk, n_rays - off);
    int rc = emit_rays(ctx, first_ray + off, m, seed);
    if (rc) return rc;
    rc = launch_trace(ctx, first_ray + off, m, seed, flags, (const double*)ctx->em_o.p,
                      (const double*)ctx->em_d.p, nullptr);
    if (rc) return rc;
  }
  return ODW_OK;
}

// ---- batches: scenes of one structure in one launch (v9) ---------------------------------------------------------
namespace {
void batch_unselect(odw_ctx* ctx) {
  if (ctx->batch_selected < 0 && !ctx->archive_selected) return;
  ctx->archive_selected = false;
  ctx->hits = ctx->own_hits;
  ctx->hit_count = ctx->own_hit_count;
  ctx->hit_capacity = ctx->own_capacity;
  ctx->hit_slots = ctx->own_slots;
  ctx->hit_ray_begin = ctx->own_ray_begin;
  ctx->hit_ray_end = ctx->own_ray_end;
  ctx->own_hits = DevBuf();
  ctx->own_hit_count = DevBuf();
  ctx->batch_selected = -1;
  ctx->ph_valid = false;
}

// the part of a scene's host tables that is STRUCTURE (what scenes of a batch must share)
bool same_structure(const odw_ctx& a, const odw_ctx& b, std::string& why) {
  if (a.P.scene.n_prims != b.P.scene.n_prims || a.P.scene.n_groups != b.P.scene.n_groups) { why = "primitive or group count"; return false; }
  for (size_t i = 0; i < a.h_prim_i32.size(); ++i) {
    const int32_t mask = (i % 4 == 2) ? ~(int32_t)ODW_FLAG_ISOLATED : ~0;       // (a matter of the boxes' values)
    if ((a.h_prim_i32[i] & mask) != (b.h_prim_i32[i] & mask)) { why = "primitive kinds, groups, flags or trimming lists"; return false; }
  }
  if (a.h_cond != b.h_cond) { why = "trimming conditions"; return false; }
  if (a.h_group_i32 != b.h_group_i32) { why = "optical types / recording switches / grating kinds"; return false; }
  if (a.h_seq != b.h_seq || a.P.scene.seq_enabled != b.P.scene.seq_enabled || a.P.scene.seq_len != b.P.scene.seq_len ||
      a.P.scene.ignore_mask != b.P.scene.ignore_mask) { why = "tracing sequence / ignored groups"; return false; }
  if (a.lean != b.lean) { why = "gratings or absorbing media in some scenes only"; return false; }
  return true;
}
}  // namespace

int odw_upload_scene_batch(odw_ctx* ctx, const odw_scene_desc* scenes, int32_t n_scenes) {
  if (!ctx || !scenes || n_scenes < 1) return fail(ctx, ODW_ERR_INVALID, "odw_upload_scene_batch: bad argument");
  if (!ctx->have_limits) return fail(ctx, ODW_ERR_NO_SCENE, "odw_upload_scene_batch before odw_set_limits (the boxes carry the tolerance)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->batch_n = 0;
  ctx->batch_traced = 0;
  ctx->batch_rows_ok = false;
  ctx->phb_valid = ctx->phb_projected = false;
  // scene 0 becomes the context's scene (shared integer tables, kernel choice, the compiled kernel's structure)
  int rc = odw_upload_scene(ctx, &scenes[0]);
  if (rc) return rc;
  if ((rc = build_bvh(ctx))) return rc;
  if (ctx->P.scene.n_nodes || ctx->P.grid.nx > 0)
    return fail(ctx, ODW_ERR_UNSUPPORTED, "odw_upload_scene_batch: batches are traced by the flat kernels (analytic scenes of up to 64 primitives)");
  const bool compiled = ctx->compile_mode != ODW_COMPILE_OFF && spec_ineligible(ctx).empty();
  const std::string text0 = compiled ? spec_text(ctx) : std::string();
  const size_t n = (size_t)ctx->P.scene.n_prims;
  // one block of doubles per scene: prim_f64 (16 n) | prim_hdr (8 n) | group_f64 (4 x 64) | group_gdir (3 x 64)
  const size_t o_hdr = 16 * n, o_gf = o_hdr + 8 * n, o_gd = o_gf + ODW_MAX_GROUPS * 4, stride = o_gd + ODW_MAX_GROUPS * 3;
  std::vector<double> blocks(stride * (size_t)n_scenes, 0.0);
  for (int k = 0; k < n_scenes; ++k) {
    odw_ctx tmp;                     // host tables only: no device, no stream
    std::memset(&tmp.P, 0, sizeof tmp.P);
    if ((rc = scene_host_tables(&tmp, &scenes[k]))) { ctx->err = tmp.err; g_error = tmp.err; return rc; }
    std::string why;
    if (!same_structure(*ctx, tmp, why))
      return fail(ctx, ODW_ERR_UNSUPPORTED, "odw_upload_scene_batch: scene " + std::to_string(k) + " differs from scene 0 in structure (" + why + ")");
    tmp.P.lim = ctx->P.lim;
    tmp.have_limits = true;
    std::vector<Box> boxes;
    std::vector<char> dead;
    compute_boxes(&tmp, boxes, dead);
    if (compiled && spec_text(&tmp) != text0)
      return fail(ctx, ODW_ERR_UNSUPPORTED, "odw_upload_scene_batch: scene " + std::to_string(k) + " differs from scene 0 in the structure a "
                                            "compiled kernel is built from (which frame entries are 0 / +1 / -1, shared boxes)");
    double* b = blocks.data() + stride * (size_t)k;
    std::memcpy(b, tmp.h_prim_f64.data(), 16 * n * sizeof(double));
    std::memcpy(b + o_hdr, tmp.h_prim_hdr.data(), 8 * n * sizeof(double));
    // (the isolated-solid shortcut depends on the boxes' values; it never changes a result: left out of batches)
    for (size_t p = 0; p < n; ++p) {
      int32_t w[4];
      std::memcpy(w, b + o_hdr + 8 * p + 6, sizeof w);
      w[2] &= ~ODW_FLAG_ISOLATED;
      std::memcpy(b + o_hdr + 8 * p + 6, w, sizeof w);
    }
    std::memcpy(b + o_gf, tmp.h_group_f64.data(), ODW_MAX_GROUPS * 4 * sizeof(double));
    std::memcpy(b + o_gd, tmp.h_group_gdir.data(), ODW_MAX_GROUPS * 3 * sizeof(double));
  }
  if ((rc = upload(ctx, ctx->batch_values, blocks.data(), blocks.size() * sizeof(double)))) return rc;
  // the shared integer tables without the isolated-solid flag
  std::vector<int32_t> pi = ctx->h_prim_i32;
  for (size_t p = 0; p < n; ++p) pi[4 * p + 2] &= ~ODW_FLAG_ISOLATED;
  if (n && (rc = upload(ctx, ctx->prim_i32, pi.data(), pi.size() * sizeof(int32_t)))) return rc;
  if ((rc = upload_done(ctx))) return rc;
  ctx->h_prim_i32 = pi;
  ctx->batch_n = n_scenes;
  ctx->batch_prims = n;
  ctx->batch_stride = stride;
  ctx->batch_spec_text = text0;
  return ODW_OK;
}

namespace {
// slots of a scene's segment: the rows asked for + the slack of block reservations
uint64_t batch_slots(odw_ctx* ctx, uint64_t rows_per_scene) {
  uint64_t slots = rows_per_scene;
  if (rows_per_scene >= kHitBlockMinRows)
    slots += hit_block_room(rows_per_scene, std::min<uint64_t>((uint64_t)ctx->n_cu * 8 * 4, std::max<uint64_t>(64, rows_per_scene / 256)), kHitBlock);
  return slots;
}
}  // namespace

// Room for batch launches of up to n_scenes scenes x rays_per_scene rays x rows_per_scene rows and for their post-hoc
// chain, allocated NOW: a buffer that has to grow in the middle of a sweep is released and allocated again, which waits for
// every stream of the device (and a hit list of 13 GB takes half a second to allocate).
int odw_batch_reserve(odw_ctx* ctx, int32_t n_scenes, uint64_t rays_per_scene, uint64_t rows_per_scene) {
  if (!ctx || n_scenes < 1 || rays_per_scene == 0) return fail(ctx, ODW_ERR_INVALID, "odw_batch_reserve: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t S = (uint64_t)n_scenes, slots = batch_slots(ctx, rows_per_scene);
  if (slots > 0x7FFFFFFFull) return fail(ctx, ODW_ERR_CAPACITY, "odw_batch_reserve: more than 2^31 rows per scene");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int rc = ODW_OK;
  if (rows_per_scene) {
    rc = ensure(ctx, ctx->batch_hits, S * slots * sizeof(odw_hit));
    if (!rc) rc = ensure(ctx, ctx->batch_hit_count, S * 4 * sizeof(uint64_t));
  }
  if (!rc) rc = phb_reserve(ctx, (int)S, rays_per_scene, slots);
  if (!rc && rows_per_scene && rays_per_scene <= (1ull << 28) && !(getenv("ODW_BATCH_PTS") && getenv("ODW_BATCH_PTS")[0] == '0'))
    rc = ensure(ctx, ctx->phb_pts, S * slots * 3 * sizeof(double));
  // (the rays generated once per launch, DeviceBatch.gen_dirs: with origins, whatever the source will be)
  if (!rc && S >= 3) rc = ensure(ctx, ctx->batch_rays_buf, (size_t)((rays_per_scene + 31) / 32 * 32 * 6 + 4) * sizeof(double));
  return rc;
}

int odw_trace_batch(odw_ctx* ctx, uint64_t first_ray, uint64_t rays_per_scene, uint64_t seed, uint32_t flags,
                    uint64_t rows_per_scene) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_trace_batch: null ctx");
  if (ctx->batch_n < 1)
    return fail(ctx, ODW_ERR_NO_SCENE, "odw_trace_batch before odw_upload_scene_batch (odw_upload_scene and a new dist_tol discard an uploaded batch)");
  // the batch's tables stand beside the single-scene tables build_bvh() writes: a rebuild now would hand the BATCH kernel
  // a one-scene box table (odw_upload_scene_batch leaves everything built)
  if (ctx->bvh_dirty || (size_t)ctx->P.scene.n_prims != ctx->batch_prims)
    return fail(ctx, ODW_ERR_NO_SCENE, "odw_trace_batch: the scene changed after odw_upload_scene_batch");
  if (ctx->emitter_active) return fail(ctx, ODW_ERR_UNSUPPORTED, "odw_trace_batch: point sources only");
  if (rays_per_scene == 0) return ODW_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  ctx->batch_rows_ok = false;            // (until this launch has been issued: a failed one leaves nothing to select)
  const uint64_t S = (uint64_t)ctx->batch_n;
  if (flags & ODW_TRACE_RECORD_HITS) {
    if (rows_per_scene == 0) return fail(ctx, ODW_ERR_CAPACITY, "odw_trace_batch: ODW_TRACE_RECORD_HITS with rows_per_scene = 0");
    const uint64_t slots = batch_slots(ctx, rows_per_scene);
    if (slots > 0x7FFFFFFFull) return fail(ctx, ODW_ERR_CAPACITY, "odw_trace_batch: more than 2^31 rows per scene");
    // (a list that has to grow is released first, which waits for the device; one that is large enough is written by
    //  this stream's next launch, behind whatever this stream still does with it)
    int rc = ensure(ctx, ctx->batch_hits, S * slots * sizeof(odw_hit));
    if (!rc) rc = ensure(ctx, ctx->batch_hit_count, S * 4 * sizeof(uint64_t));
    if (rc) return rc;
    ctx->batch_seg_slots = slots;
    ctx->batch_seg_capacity = rows_per_scene;
    HIPCHK(ctx, hipMemsetAsync(ctx->batch_hit_count.p, 0, S * 4 * sizeof(uint64_t), ctx->stream));
    // every recorded row notes its slot at its ray's place (DeviceOutputs.row_of): the table starts out as "no row"
    ctx->batch_marked = false;
    ctx->batch_pts = false;
    if (rays_per_scene <= (1ull << 28)) {
      const uint64_t rays_pad = (rays_per_scene + 31) / 32 * 32;
      rc = ensure(ctx, ctx->phb_row_of, S * rays_pad * sizeof(uint32_t));
      if (rc) return rc;
      HIPCHK(ctx, hipMemsetAsync(ctx->phb_row_of.p, 0xff, S * rays_pad * sizeof(uint32_t), ctx->stream));
      ctx->batch_marked = true;
      // (the points alone, by slot: 24 bytes more per row; ODW_BATCH_PTS=0: the projection reads the rows)
      static const bool pts_off = [] { const char* e = getenv("ODW_BATCH_PTS"); return e && e[0] == '0'; }();
      ctx->batch_pts = !pts_off && ensure(ctx, ctx->phb_pts, S * slots * 3 * sizeof(double)) == ODW_OK;
    }
  }
  // the value tables of scene 0 stand where the kernels' pointers point; scene s lies s strides further
  const size_t n = (size_t)ctx->P.scene.n_prims;
  const double* base = (const double*)ctx->batch_values.p;
  DeviceScene saved = ctx->P.scene;
  ctx->P.scene.prim_f64 = base;
  ctx->P.scene.prim_hdr = base + 16 * n;
  ctx->P.scene.group_f64 = base + 24 * n;
  ctx->P.scene.group_gdir = base + 24 * n + ODW_MAX_GROUPS * 4;
  ctx->batch_launch = true;
  ctx->phb_valid = ctx->phb_projected = false;
  ctx->batch_traced = ctx->batch_n;
  ctx->batch_rays = rays_per_scene;
  ctx->batch_first = first_ray;
  const int rc = launch_trace(ctx, first_ray, rays_per_scene, seed, flags, nullptr, nullptr, nullptr);
  ctx->batch_launch = false;
  ctx->batch_rows_ok = rc == ODW_OK && (flags & ODW_TRACE_RECORD_HITS) != 0;
  if (rc) ctx->batch_traced = 0;
  const bool dirty = ctx->bvh_dirty;     // (launch_trace may have rebuilt boxes: keep what it set, restore the pointers only)
  (void)dirty;
  ctx->P.scene.prim_f64 = saved.prim_f64;
  ctx->P.scene.prim_hdr = saved.prim_hdr;
  ctx->P.scene.group_f64 = saved.group_f64;
  ctx->P.scene.group_gdir = saved.group_gdir;
  return rc;
}

int odw_batch_select(odw_ctx* ctx, int32_t scene) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_batch_select: null ctx");
  if (scene < 0 || ctx->archive_selected) batch_unselect(ctx);
  if (scene < 0) return ODW_OK;
  if (!ctx->batch_rows_ok || scene >= ctx->batch_traced || !ctx->batch_hits.p || !ctx->batch_seg_slots)
    return fail(ctx, ODW_ERR_INVALID, "odw_batch_select: no such segment (odw_trace_batch with ODW_TRACE_RECORD_HITS first)");
  if (ctx->batch_selected < 0) {
    ctx->own_hits = ctx->hits;
    ctx->own_hit_count = ctx->hit_count;
    ctx->own_capacity = ctx->hit_capacity;
    ctx->own_slots = ctx->hit_slots;
    ctx->own_ray_begin = ctx->hit_ray_begin;
    ctx->own_ray_end = ctx->hit_ray_end;
  }
  ctx->hits.p = (odw_hit*)ctx->batch_hits.p + (size_t)scene * ctx->batch_seg_slots;
  ctx->hits.bytes = ctx->batch_seg_slots * sizeof(odw_hit);
  ctx->hit_count.p = (uint64_t*)ctx->batch_hit_count.p + 4 * (size_t)scene;
  ctx->hit_count.bytes = 2 * sizeof(uint64_t);
  ctx->hit_capacity = ctx->batch_seg_capacity;
  ctx->hit_slots = ctx->batch_seg_slots;
  ctx->hit_ray_begin = ctx->batch_first;
  ctx->hit_ray_end = std::min<uint64_t>(ctx->batch_first + ctx->batch_rays, 1ull << 48);
  ctx->batch_selected = scene;
  ctx->ph_valid = false;
  return ODW_OK;
}

int odw_batch_rows(odw_ctx* ctx, uint64_t* rows, uint64_t* wanted, int32_t n) {
  if (!ctx || !rows || n < 0) return fail(ctx, ODW_ERR_INVALID, "odw_batch_rows: bad argument");
  if (n > 0 && !ctx->batch_rows_ok) return fail(ctx, ODW_ERR_INVALID, "odw_batch_rows: no batch was traced with hit rows");
  if (n > ctx->batch_traced) return fail(ctx, ODW_ERR_INVALID, "odw_batch_rows: more scenes than the batch traced");
  if (n == 0) return ODW_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<uint64_t> v(4 * (size_t)n, 0);
  HIPCHK(ctx, hipMemcpyAsync(v.data(), ctx->batch_hit_count.p, v.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < n; ++k) {
    const uint64_t used = std::min<uint64_t>(v[4 * k], ctx->batch_seg_slots);
    rows[k] = used > v[4 * k + 1] ? used - v[4 * k + 1] : 0;
    if (wanted) wanted[k] = v[4 * k];          // slots asked for (above the segment's room: rows were dropped)
  }
  return ODW_OK;
}

static int hit_slots_used(odw_ctx* ctx, uint64_t* used, uint64_t* rows);

// ---- a run's rows kept in HBM (v9) ---------------------------------------------------------------------------------
int odw_archive_append(odw_ctx* ctx, odw_ctx* src, uint64_t* total_rows) {
  if (!ctx || !src) return fail(ctx, ODW_ERR_INVALID, "odw_archive_append: null context");
  if (ctx->device != src->device) return fail(ctx, ODW_ERR_INVALID, "odw_archive_append: the contexts live on different devices");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  if (src != ctx) batch_unselect(src);
  uint64_t used = 0, rows = 0;
  int rc = hit_slots_used(src, &used, &rows);          // (waits for src's stream: its launch has finished)
  if (rc) { ctx->err = src->err; return rc; }
  if (used) {
    const uint64_t need = ctx->archive_slots + used;
    if (need > 0x7FFFFFFFull) return fail(ctx, ODW_ERR_CAPACITY, "odw_archive_append: more than 2^31 rows");
    if (ctx->archive.bytes < need * sizeof(odw_hit)) {
      // grow by doubling: the rows kept so far move once per doubling
      DevBuf bigger;
      const uint64_t cap = std::max<uint64_t>(need, std::max<uint64_t>(1ull << 22, 2 * ctx->archive.bytes / sizeof(odw_hit)));
      HIPCHK(ctx, hipMalloc(&bigger.p, cap * sizeof(odw_hit)));
      bigger.bytes = cap * sizeof(odw_hit);
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->archive_slots)
        HIPCHK(ctx, hipMemcpyAsync(bigger.p, ctx->archive.p, ctx->archive_slots * sizeof(odw_hit), hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      release(ctx->archive);
      ctx->archive = bigger;
    }
    // (slots tagged unused travel along: every pass over a hit list skips them)
    HIPCHK(ctx, hipMemcpyAsync((odw_hit*)ctx->archive.p + ctx->archive_slots, src->hits.p, used * sizeof(odw_hit),
                               hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));    // (src's list may be recycled as soon as this returns)
    ctx->archive_ray_begin = ctx->archive_slots ? std::min(ctx->archive_ray_begin, src->hit_ray_begin) : src->hit_ray_begin;
    ctx->archive_ray_end = ctx->archive_slots ? std::max(ctx->archive_ray_end, src->hit_ray_end) : src->hit_ray_end;
    ctx->archive_slots = need;
    ctx->archive_unused += used - rows;
  }
  if (total_rows) *total_rows = ctx->archive_slots - ctx->archive_unused;
  return ODW_OK;
}

int odw_archive_select(odw_ctx* ctx, int32_t on) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_archive_select: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  if (!on) return ODW_OK;
  if (!ctx->archive_slots) return fail(ctx, ODW_ERR_INVALID, "odw_archive_select: nothing was archived");
  int rc = ensure(ctx, ctx->archive_count, 2 * sizeof(uint64_t));
  if (rc) return rc;
  const uint64_t count[2] = {ctx->archive_slots, ctx->archive_unused};
  HIPCHK(ctx, hipMemcpyAsync(ctx->archive_count.p, count, sizeof count, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->own_hits = ctx->hits;
  ctx->own_hit_count = ctx->hit_count;
  ctx->own_capacity = ctx->hit_capacity;
  ctx->own_slots = ctx->hit_slots;
  ctx->own_ray_begin = ctx->hit_ray_begin;
  ctx->own_ray_end = ctx->hit_ray_end;
  ctx->hits.p = ctx->archive.p;
  ctx->hits.bytes = ctx->archive_slots * sizeof(odw_hit);
  ctx->hit_count.p = ctx->archive_count.p;
  ctx->hit_count.bytes = 2 * sizeof(uint64_t);
  ctx->hit_capacity = ctx->hit_slots = ctx->archive_slots;
  ctx->hit_ray_begin = ctx->archive_ray_begin;
  ctx->hit_ray_end = ctx->archive_ray_end;
  ctx->archive_selected = true;
  ctx->ph_valid = false;
  return ODW_OK;
}

int odw_archive_reset(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_archive_reset: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  release(ctx->archive);
  ctx->archive_slots = ctx->archive_unused = 0;
  return ODW_OK;
}

int odw_trace_rays(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, const double* origins,
                   const double* directions, const double* powers, uint32_t flags) {
  if (!ctx || !origins || !directions) return fail(ctx, ODW_ERR_INVALID, "odw_trace_rays: null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  if (n_rays == 0) return ODW_OK;
  // the previous launch may still read the staging buffers
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int rc;
  // the caller's n x 3 arrays are staged as they are, then turned component-major on the device (the kernels'
  // lanes read consecutive rays: unit-stride loads instead of 24-byte strides)
  if ((rc = ensure(ctx, ctx->ray_aos, n_rays * 6 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->ray_o, n_rays * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->ray_d, n_rays * 3 * sizeof(double)))) return rc;
  double* aos = (double*)ctx->ray_aos.p;
  HIPCHK(ctx, hipMemcpyAsync(aos, origins, n_rays * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(aos + 3 * n_rays, directions, n_rays * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    const unsigned tgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((3 * n_rays + 255) / 256, (uint64_t)ctx->n_cu * 16));
    hipLaunchKernelGGL(odw_rays_to_components_kernel, dim3(tgrid), dim3(256), 0, ctx->stream, (const double*)aos,
                       (const double*)(aos + 3 * n_rays), n_rays, (double*)ctx->ray_o.p, (double*)ctx->ray_d.p);
    HIPCHK(ctx, hipGetLastError());
  }
  if (powers) {
    if ((rc = upload(ctx, ctx->ray_p, powers, n_rays * sizeof(double)))) return rc;
  }
  rc = launch_trace(ctx, first_ray, n_rays, ctx->surface_seed, flags, (const double*)ctx->ray_o.p,
                    (const double*)ctx->ray_d.p, powers ? (const double*)ctx->ray_p.p : nullptr);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // caller's arrays may go away
  return ODW_OK;
}

int odw_sync(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_sync: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

int odw_reset_results(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_reset_results: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  batch_unselect(ctx);
  ctx->ph_valid = false;
  ctx->hit_ray_end = 0;
  HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, ODW_CNT_COUNT * sizeof(uint64_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->hit_count.p, 0, 2 * sizeof(uint64_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->seg_count.p, 0, sizeof(uint64_t), ctx->stream));
  if (ctx->n_bins) HIPCHK(ctx, hipMemsetAsync(ctx->hist.p, 0, ctx->n_bins * sizeof(uint64_t), ctx->stream));
  return ODW_OK;
}

int odw_reset_segments(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_reset_segments: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemsetAsync(ctx->seg_count.p, 0, sizeof(uint64_t), ctx->stream));
  return ODW_OK;
}

int odw_reset_hits(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_reset_hits: null ctx");
  batch_unselect(ctx);
  ctx->ph_valid = false;
  ctx->hit_ray_end = 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemsetAsync(ctx->hit_count.p, 0, 2 * sizeof(uint64_t), ctx->stream));
  return ODW_OK;
}

int odw_fetch_counters(odw_ctx* ctx, uint64_t* out, int32_t n) {
  if (!ctx || !out || n < 0) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_counters: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t tmp[ODW_CNT_COUNT];
  HIPCHK(ctx, hipMemcpyAsync(tmp, ctx->counters.p, sizeof tmp, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n && i < ODW_CNT_COUNT; ++i) out[i] = tmp[i];
  return ODW_OK;
}

// slots handed out (clamped to the buffer) and how many of them hold rows
static int hit_slots_used(odw_ctx* ctx, uint64_t* used, uint64_t* rows) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t v[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(v, ctx->hit_count.p, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *used = std::min<uint64_t>(v[0], ctx->hit_slots);
  *rows = *used > v[1] ? *used - v[1] : 0;
  return ODW_OK;
}

int odw_hit_count(odw_ctx* ctx, uint64_t* n) {
  if (!ctx || !n) return fail(ctx, ODW_ERR_INVALID, "odw_hit_count: bad argument");
  uint64_t used = 0;
  return hit_slots_used(ctx, &used, n);
}

int odw_fetch_hits(odw_ctx* ctx, odw_hit* out, uint64_t capacity, uint64_t* n) {
  if (!ctx || !n) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_hits: bad argument");
  uint64_t used = 0, have = 0;
  int rc = hit_slots_used(ctx, &used, &have);
  if (rc) return rc;
  *n = have;
  if (!out || capacity == 0) return ODW_OK;
  if (have > capacity) return fail(ctx, ODW_ERR_CAPACITY, "odw_fetch_hits: output buffer too small");
  ctx->ph_valid = false;           // the sort buffers are shared with odw_hits_select
  if (have) {
    // append order is scheduling dependent; a ray's own rows are appended in
    // bounce order, so a STABLE sort by ray index gives (ray, bounce) order.
    // Done on the device: LSD radix sort of (ray index -> slot number) pairs
    // (hipCUB, stable; unused slots of block reservations sort to the end),
    // then a gather of the 64-byte rows, then one D2H copy.
    if (used > 0x7FFFFFFFull) return fail(ctx, ODW_ERR_CAPACITY, "odw_fetch_hits: more than 2^31 rows per fetch");
    for (int k = 0; k < 2; ++k) {
      if ((rc = ensure(ctx, ctx->sort_keys[k], used * sizeof(uint64_t)))) return rc;
      if ((rc = ensure(ctx, ctx->sort_vals[k], used * sizeof(uint32_t)))) return rc;
    }
    if ((rc = ensure(ctx, ctx->sorted_rows, have * sizeof(odw_hit)))) return rc;
    uint64_t* k_in = (uint64_t*)ctx->sort_keys[0].p;
    uint64_t* k_out = (uint64_t*)ctx->sort_keys[1].p;
    uint32_t* v_in = (uint32_t*)ctx->sort_vals[0].p;
    uint32_t* v_out = (uint32_t*)ctx->sort_vals[1].p;
    const unsigned blocks = (unsigned)((used + 255) / 256);
    // (only the bits ray indices of this list can have, + 1 for the sentinel: see odw_hits_select)
    int bits = 48;
    if (ctx->hit_ray_end && ctx->hit_ray_end < (1ull << 48)) { bits = 1; while ((1ull << bits) < ctx->hit_ray_end) ++bits; }
    const bool spare = ctx->hit_ray_end && ctx->hit_ray_end < (1ull << bits);
    const uint64_t sentinel = spare ? (1ull << bits) - 1 : 1ull << bits;
    if (spare) --bits;
    hipLaunchKernelGGL(hit_keys_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p,
                       used, sentinel, k_in, v_in);
    HIPCHK(ctx, hipGetLastError());
    size_t tmp_bytes = 0;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (int)used, 0, bits + 1,
                                                   ctx->stream));
    if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (int)used,
                                                   0, bits + 1, ctx->stream));
    const unsigned gblocks = (unsigned)((have * 4 + 255) / 256);
    hipLaunchKernelGGL(hit_gather_kernel, dim3(gblocks), dim3(256), 0, ctx->stream, (const odw_hit*)ctx->hits.p,
                       v_out, have, (odw_hit*)ctx->sorted_rows.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->sorted_rows.p, have * sizeof(odw_hit), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return ODW_OK;
}

int odw_swap_hit_lists(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_swap_hit_lists: null ctx");
  batch_unselect(ctx);
  if (ctx->hit_capacity == 0) return fail(ctx, ODW_ERR_CAPACITY, "odw_swap_hit_lists without odw_reserve_hits");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->ph_valid = false;
  if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!ctx->alt_ready) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->alt_ready, hipEventDisableTiming));
  if (ctx->alt_slots < ctx->hit_slots) {     // the other list gets the same room
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    release(ctx->alt_hits);
    int rc = ensure(ctx, ctx->alt_hits, ctx->hit_slots * sizeof(odw_hit));
    if (!rc) rc = ensure(ctx, ctx->alt_hit_count, 2 * sizeof(uint64_t));
    if (rc) return rc;
    ctx->alt_capacity = ctx->hit_capacity;
    ctx->alt_slots = ctx->hit_slots;
    HIPCHK(ctx, hipMemsetAsync(ctx->alt_hit_count.p, 0, 2 * sizeof(uint64_t), ctx->stream));
  }
  // everything launched so far wrote into the list that is put aside now
  HIPCHK(ctx, hipEventRecord(ctx->alt_ready, ctx->stream));
  std::swap(ctx->hits, ctx->alt_hits);
  std::swap(ctx->hit_count, ctx->alt_hit_count);
  std::swap(ctx->hit_capacity, ctx->alt_capacity);
  std::swap(ctx->hit_slots, ctx->alt_slots);
  std::swap(ctx->hit_ray_end, ctx->alt_hit_ray_end);
  std::swap(ctx->hit_ray_begin, ctx->alt_hit_ray_begin);
  ctx->swapping = true;
  return ODW_OK;
}

int odw_host_alloc(odw_ctx* ctx, uint64_t bytes, void** out) {
  if (!ctx || !out || bytes == 0) return fail(ctx, ODW_ERR_INVALID, "odw_host_alloc: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  *out = nullptr;
  HIPCHK(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return ODW_OK;
}

int odw_mem_info(odw_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_mem_info: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t f = 0, t = 0;
  HIPCHK(ctx, hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (uint64_t)f;
  if (total_bytes) *total_bytes = (uint64_t)t;
  return ODW_OK;
}

int odw_host_free(odw_ctx* ctx, void* p) {
  // (ctx may be null: page-locked arrays handed to the caller can outlive the context that allocated them)
  if (p) HIPCHK(ctx, hipHostFree(p));
  return ODW_OK;
}

int odw_release_swapped_hits(odw_ctx* ctx) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_release_swapped_hits: null ctx");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  release(ctx->alt_hits);
  ctx->alt_capacity = ctx->alt_slots = 0;
  ctx->swapping = false;           // launches reserve hit-list blocks again
  return ODW_OK;
}

int odw_fetch_swapped_hits(odw_ctx* ctx, odw_hit* out, uint64_t capacity, uint64_t* n) {
  if (!ctx || !n) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_swapped_hits: bad argument");
  if (!ctx->swapping || !ctx->copy_stream) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_swapped_hits: odw_swap_hit_lists first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // the copy stream waits for the launches that filled the list, not for what runs on the trace
  // stream since: the next launch proceeds while these rows cross PCIe
  HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->alt_ready, 0));
  uint64_t v[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(v, ctx->alt_hit_count.p, sizeof v, hipMemcpyDeviceToHost, ctx->copy_stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
  const uint64_t have = std::min<uint64_t>(v[0], ctx->alt_slots);    // dense: no unused slots in swapped lists
  *n = have;
  if (!out || capacity == 0 || have == 0) return ODW_OK;
  if (have > capacity) return fail(ctx, ODW_ERR_CAPACITY, "odw_fetch_swapped_hits: output buffer too small");
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->alt_hits.p, have * sizeof(odw_hit), hipMemcpyDeviceToHost, ctx->copy_stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
  return ODW_OK;
}

int odw_segment_count(odw_ctx* ctx, uint64_t* n, uint64_t* dropped) {
  if (!ctx || !n) return fail(ctx, ODW_ERR_INVALID, "odw_segment_count: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t wanted = 0;
  HIPCHK(ctx, hipMemcpyAsync(&wanted, ctx->seg_count.p, sizeof wanted, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *n = std::min<uint64_t>(wanted, ctx->seg_capacity);
  if (dropped) *dropped = wanted - *n;
  return ODW_OK;
}

int odw_fetch_segments(odw_ctx* ctx, odw_segment* out, uint64_t capacity, uint64_t* n) {
  if (!ctx || !n) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_segments: bad argument");
  ctx->ph_valid = false;           // the sort buffers are shared with odw_hits_select
  uint64_t have = 0;
  int rc = odw_segment_count(ctx, &have, nullptr);
  if (rc) return rc;
  *n = have;
  if (!out || capacity == 0 || have == 0) return ODW_OK;
  if (have > capacity) return fail(ctx, ODW_ERR_CAPACITY, "odw_fetch_segments: output buffer too small");
  // same device-side ordering as the hit list, with the explicit key (ray, ordinal)
  static_assert(sizeof(odw_segment) == sizeof(odw_hit), "rows share the gather kernel");
  for (int k = 0; k < 2; ++k) {
    if ((rc = ensure(ctx, ctx->sort_keys[k], have * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals[k], have * sizeof(uint32_t)))) return rc;
  }
  if ((rc = ensure(ctx, ctx->sorted_rows, have * sizeof(odw_segment)))) return rc;
  uint64_t* k_in = (uint64_t*)ctx->sort_keys[0].p;
  uint64_t* k_out = (uint64_t*)ctx->sort_keys[1].p;
  uint32_t* v_in = (uint32_t*)ctx->sort_vals[0].p;
  uint32_t* v_out = (uint32_t*)ctx->sort_vals[1].p;
  hipLaunchKernelGGL(seg_keys_kernel, dim3((unsigned)((have + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const odw_segment*)ctx->segs.p, have, k_in, v_in);
  HIPCHK(ctx, hipGetLastError());
  size_t tmp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (int)have, 0, 52,
                                                 ctx->stream));
  if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes))) return rc;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (int)have, 0,
                                                 52, ctx->stream));
  hipLaunchKernelGGL(hit_gather_kernel, dim3((unsigned)((have * 4 + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const odw_hit*)ctx->segs.p, v_out, have, (odw_hit*)ctx->sorted_rows.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->sorted_rows.p, have * sizeof(odw_segment), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

int odw_fetch_histogram(odw_ctx* ctx, uint64_t* out, uint64_t n_bins) {
  if (!ctx || !out) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_histogram: bad argument");
  if (n_bins != ctx->n_bins || n_bins == 0) return fail(ctx, ODW_ERR_INVALID, "odw_fetch_histogram: bin count mismatch");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->hist.p, n_bins * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

int odw_sample(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, uint64_t seed, double* theta_out,
               double* phi_out) {
  if (!ctx || !theta_out || !phi_out) return fail(ctx, ODW_ERR_INVALID, "odw_sample: bad argument");
  if (!ctx->have_source || ctx->emitter_active) return fail(ctx, ODW_ERR_NO_SCENE, "point source not uploaded");
  if (n_rays == 0) return ODW_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->samp_t, n_rays * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->samp_phi, n_rays * sizeof(double)))) return rc;
  const unsigned grid = (unsigned)std::min<uint64_t>((n_rays + 255) / 256, (uint64_t)ctx->n_cu * 8);
  hipLaunchKernelGGL(odw_sample_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->P.source, first_ray,
                     n_rays, seed, (double*)ctx->samp_t.p, (double*)ctx->samp_phi.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(theta_out, ctx->samp_t.p, n_rays * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(phi_out, ctx->samp_phi.p, n_rays * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return ODW_OK;
}

int odw_device_results(odw_ctx* ctx, void** dptr, uint64_t* n_words, uint64_t* hist_offset_words) {
  if (!ctx || !dptr || !n_words || !hist_offset_words) return fail(ctx, ODW_ERR_INVALID, "odw_device_results: bad argument");
  *dptr = ctx->results.p;
  *n_words = kResultsHead + ctx->n_bins;
  *hist_offset_words = kResultsHead;
  return ODW_OK;
}

int odw_device_histogram(odw_ctx* ctx, void** dptr, uint64_t* n_bins) {
  if (!ctx || !dptr || !n_bins) return fail(ctx, ODW_ERR_INVALID, "odw_device_histogram: bad argument");
  *dptr = ctx->n_bins ? ctx->hist.p : nullptr;
  *n_bins = ctx->n_bins;
  return ODW_OK;
}

int odw_device_counters(odw_ctx* ctx, void** dptr, uint64_t* n) {
  if (!ctx || !dptr || !n) return fail(ctx, ODW_ERR_INVALID, "odw_device_counters: bad argument");
  *dptr = ctx->counters.p;
  *n = ODW_CNT_COUNT;
  return ODW_OK;
}

int odw_stream(odw_ctx* ctx, void** hip_stream) {
  if (!ctx || !hip_stream) return fail(ctx, ODW_ERR_INVALID, "odw_stream: bad argument");
  *hip_stream = (void*)ctx->stream;
  return ODW_OK;
}

int odw_timing_enable(odw_ctx* ctx, int on) {
  if (!ctx) return fail(ctx, ODW_ERR_INVALID, "odw_timing_enable: null ctx");
  ctx->timing = on != 0;
  return ODW_OK;
}

int odw_timing_read(odw_ctx* ctx, double* total_ms, uint64_t* launches) {
  if (!ctx || !total_ms || !launches) return fail(ctx, ODW_ERR_INVALID, "odw_timing_read: bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& ev : ctx->events) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ev.first, ev.second));
    ctx->timing_ms += ms;
    ctx->timing_launches += 1;
    ctx->free_events.push_back(ev);
  }
  ctx->events.clear();
  *total_ms = ctx->timing_ms;
  *launches = ctx->timing_launches;
  ctx->timing_ms = 0;
  ctx->timing_launches = 0;
  return ODW_OK;
}

}  // extern "C"

#include "odw_posthoc.hip"
#include "odw_posthoc_batch.hip"
