// odw_device.h -- device-side tables and math of the gfx950 ray-tracing core.
//
// Layout in HBM (all read-only during a launch, L2/scalar-cache resident):
//   prim_f64 : n_prims x 16 f64   rows 0..11 global->local (R|t), 12..15 params
//              (TRIANGLE: v0, e1 = v1-v0, e2 = v2-v0, unit facet normal | barycentric slack per unit of
//               tolerance for u, v, u+v; unused)
//   prim_hdr : n_prims x 64 B     global bounding box (6 f64) + type, group, flags, conds (4 i32)
//   prim_i32 : n_prims x 4  i32   type, group, flags|facemask<<8, cond_off|cnt<<24
//   cond_i32 : n_conds      i32   prim | inside<<31
//   group_f64: 64 x 4 f64         ior, reflectivity, absorption length, grating lpm
//   group_i32: 64 x 4 i32         optical type, record, grating type, grating order
//   cdf tables: interleaved (cdf, edge) f64 pairs per knot, one 16-B load each
// Everything a wave needs per primitive is addressed with wave-uniform
// indices, so hipcc emits scalar (s_load) loads: the scene costs SGPRs, not
// VGPRs, and no LDS staging is needed for the small benchmark scenes.
#pragma once
#ifndef __HIPCC_RTC__      // (hiprtc brings the HIP runtime declarations and the fixed-width integers itself)
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef signed int int32_t;
typedef unsigned int uint32_t;
typedef signed long int64_t;
typedef unsigned long uint64_t;
typedef unsigned long uintptr_t;
#endif

#ifdef __HIPCC_RTC__        // (compiled at run time from the sources embedded in the library: flat names)
#include "odw_trace.h"
#else
#include "../../../include/odw_trace.h"
#endif

#ifndef INFINITY          // (runtime compilation: no <math.h>)
#define INFINITY (__builtin_inff())
#endif

namespace odw {

#define ODW_TAG_UNUSED 0xFFFFFFFFFFFFFFFFull   // tag of a hit-list slot that was reserved but not written
#define ODW_CNT_LDS 8        // counters 0..7 are gathered in LDS; rarer ones are added to the global array directly
#define ODW_SOLID_SHIFT 16   // prim_i32 flags word: flags | facemask << 8 | solid id << 16
// set by the library (compute_boxes): the box of the primitive's solid, tolerance slack included, keeps more than
// 4 distTol away from the box of every other solid.  A ray that has just ENTERED such a solid meets the solid
// itself before anything else (a straight line that has left a box does not come back, and every hit the
// tolerance rules accept lies in its solid's box): the next segment tests this solid's primitives only, and
// falls back to all of them in the one case that finds nothing (a hit within distTol beyond an edge).
#define ODW_FLAG_ISOLATED 0x4

struct d3 {
  double x, y, z;
};
__device__ __forceinline__ d3 mk(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct DeviceScene {
  const double* prim_f64;
  const double* prim_hdr;       // [n_prims*8] 64-B header: global AABB lo xyz, hi xyz (tolerance slack
                                //   included) | i32 type, group, flags, cond_off|cnt<<24  -- one s_load_dwordx16
  const int32_t* prim_i32;
  const int32_t* cond_i32;
  const double* group_f64;      // [64*4]
  const int32_t* group_i32;     // [64*4]
  const double* group_gdir;     // [64*3]
  const uint64_t* seq_mask;     // [seq_len]
  // BVH (big scenes only): 64-byte nodes = f32 boxes of both children
  // (lo0 hi0 lo1 hi1, 12 floats) + child0, child1, count0, count1 (count > 0: leaf)
  const float* bvh_nodes;       // [n_nodes*16]
  const int32_t* bvh_prims;     // leaf primitive order
  const float* bvh_leaf;        // 64-byte leaf records of the mesh kernel (odw_mesh.hip), or null
  const uint32_t* bvh_wide;     // its eight-wide tree: 128-byte nodes (odw_capi.hip: WideBvh)
  double wide_lo[3], wide_hi[3];   // the box of that tree's root (rays start their float32 walk where they enter it)
  const double* tri_nrm;        // [n_prims*9] vertex normals of TRIANGLE primitives, or null (facet normals)
  int32_t n_prims, n_groups, n_nodes;
  int32_t seq_enabled, seq_len;
  uint64_t ignore_mask, all_mask;
};

// Rectilinear grid over the primitives of a big analytic scene (no facets): cell (i, j, k) spans
// [bx[i], bx[i+1]] x [by[j], by[j+1]] x [bz[k], bz[k+1]]; the planes sit in the gaps between the
// primitives' boxes where there are gaps (a Draft array of spheres gets one sphere per cell), wide
// slabs are cut evenly.  A cell lists every primitive whose box (tolerance slack included) meets it.
// Walked by a 3-D DDA: no stack, one cell per step.
struct DeviceGrid {
  const double* bounds;         // [nx+1 | ny+1 | nz+1]
  const uint32_t* cells;        // [nx*ny*nz] first item | count << 24, x fastest
  const void* items;            // spheres: 48-byte records (cx, cy, cz, R, {prim, group | solid << 8}, flag word)
                                // else   : u32 primitive indices
  int32_t nx, ny, nz, n_items;
  int32_t spheres;              // every listed primitive is an untrimmed sphere
  int32_t in_lds;               // cells + items are staged in LDS by every block
  uint32_t lds_bytes;           // dynamic LDS of a block
};

struct DeviceSource {
  double m[12];                 // local -> global
  double focal_length, wavelength, power;
  const double* phi_tab;        // [n_phi_knots*2] (cdf, edge)
  const double* t_tab;          // [rows*n_t_knots*2]
  const int32_t* t_guide;       // [rows*(GUIDE+1)] bracket guide for the inverse CDF
  const int32_t* phi_guide;     // [n_phi_guide+1] the same for the azimuth table
  int32_t n_phi_knots, n_t_knots, n_t_rows, n_guide, n_phi_guide;
  int32_t finite_focal;
};

// one stochastic-surface sampler (odw_surface_sampler_desc): a family of
// source-like tables, member k = one value of the per-hit constant
struct DeviceSurfaceSampler {
  const double* phi_tab;        // [n_family][n_phi_knots*2]
  const double* t_tab;          // [n_family][rows*n_t_knots*2]
  const int32_t* t_guide;       // [n_family][rows*(n_guide+1)]
  int32_t n_phi_knots, n_t_knots, n_t_rows, n_guide;
  int32_t axis, n_family;
  double lo, inv_step;          // member = rint((c - lo) * inv_step), clamped
  double mu;                    // n1 / n2 this sampler is for (-1: total reflection, 0: every hit)
  int32_t next;                 // the next sampler of the same (group, kind), or -1
  int32_t n_atoms;              // discrete events (odw_surface_sampler_desc)
  const double* atom_mass;      // [n_family][n_atoms]
  double atom_theta[ODW_SURF_MAX_ATOMS][3], atom_phi[ODW_SURF_MAX_ATOMS][3];
};

// surface source (odw_surface_source_desc)
struct DeviceEmitter {
  const double* prim_f64;       // [n_prims*16] global->local rows, params
  const int32_t* prim_i32;      // [n_prims*4] type, flags, cond_off, cond_cnt
  const int32_t* cond_i32;      // prim | inside<<31
  const int32_t* face_i32;      // [n_faces*2] prim, face
  const double* face_cdf;       // [n_faces+1] cumulative untrimmed area / total
  const double* t_tab;          // [n_t_knots*2] (cdf, edge)
  const int32_t* t_guide;       // [n_guide+1]
  const double* tri_nrm;        // [n_prims*9] vertex normals of TRIANGLE primitives, or null
  int32_t n_faces, n_t_knots, n_guide;
  double dist_tol, wavelength, power;
};

struct DeviceLimits {
  double max_ray_length, dist_tol, power_tol;
  int32_t max_intersections;
};

struct DeviceDetector {
  double origin[3], ex[3], ey[3];
  double x_lo, y_lo, x_scale, y_scale;  // scale = n/(hi-lo)
  double nx_f, ny_f;
  int32_t nx, ny, group, enabled;
};

struct DeviceOutputs {
  odw_hit* hits;
  uint64_t hit_capacity;
  unsigned long long* hit_count;       // [0] slots handed out, [1] of these: unused (marked with ODW_TAG_UNUSED)
  uint32_t hit_block;                  // 0: one reservation per append; else slots per reservation of a wave
  unsigned long long* hist;            // nx*ny
  unsigned long long* counters;        // ODW_CNT_COUNT
  unsigned long long* chunk_counter;   // next unassigned chunk of this launch (zeroed per launch)
  odw_segment* segs;                   // ODW_TRACE_RECORD_SEGMENTS (RecordRays sources)
  uint64_t seg_capacity;
  unsigned long long* seg_count;       // rows wanted so far (may exceed the capacity)
  // batch launches (DeviceBatch): hit_count holds FOUR words per scene -- slots handed out, unused slots, rows of
  // leaving rays, spare --, and with row_of every recorded row notes its slot at its ray's place in the scene's table
  // (row_of[scene * row_stride + ray - first_ray], preset to "none" by the host): the ordered selection of the post-hoc
  // binning (odw_posthoc.hip: ph_mark_kernel) without its pass over the rows
  uint32_t* row_of;
  uint64_t row_stride;
  // ... and, with pts, writes its point once more into a table of points alone (component c at pts[(scene * 3 + c) * hit_capacity + slot]):
  // the projection of the post-hoc chain then reads 24 bytes per row instead of the 64-byte row (odw_posthoc_batch.hip)
  double* pts;
};

// A batch launch (odw_trace_batch): n_scenes scenes of ONE structure -- the same primitives, trimming lists, groups and
// optical types, different numbers (a parameter sweep: examples/1-getting-started/optimize-spotsize.ipynb cell 9) --
// traced by one grid.  The float64 tables of scene s (prim_f64, prim_hdr, group_f64, group_gdir) lie `stride` doubles
// behind those of scene s - 1; hand-out unit g of the launch belongs to scene g / chunks_per_scene, whose rays are
// first_ray ... first_ray + rays - 1 exactly as in a launch of that scene alone; its rows go to its own segment of the
// hit list (hit_capacity slots each, a pair of counters each).  A wave works on one scene at a time.
struct DeviceBatch {
  uint64_t rays;                // rays per scene
  uint64_t stride;              // doubles between the value tables of consecutive scenes
  uint32_t chunks_per_scene;
  uint32_t n_scenes;            // 0: not a batch launch
  // every scene of a batch traces the SAME rays (one source, one seed, the same ray numbers): their initial conditions
  // are generated once, by a pass before the launch, and read here -- component-major directions (x of every ray, then
  // y, z: gen_stride doubles apart), then three doubles of the common origin (a source at its focus); gen_origins: the
  // origins the same way where they differ from ray to ray.  Null: every scene generates its rays itself.
  const double* gen_dirs;
  const double* gen_origins;
  uint64_t gen_stride;
};

// Kernel argument.  The source and detector blocks live in device memory and
// are read where they are used (once per ray) through an opaque pointer:
// passed by value, hipcc hoists every field (and values derived from them)
// out of the ray loop and keeps ~40 VGPRs occupied for the whole kernel.
struct TraceParams {
  DeviceScene scene;
  DeviceGrid grid;              // grid kernels only (nx = 0: none)
  DeviceLimits lim;
  const DeviceSource* source;
  const DeviceDetector* det;
  int32_t det_enabled;
  double wavelength;            // nm; of the uploaded source, 500 if none (explicit rays)
  const DeviceSurfaceSampler* samplers;   // stochastic surfaces (STOCH kernels only)
  const int32_t* group_sampler;           // [64*2] sampler index of (group, kind) or -1
  DeviceOutputs out;
  const double* ray_origins;    // explicit initial conditions (or null), component-major: x of every ray, then
  const double* ray_dirs;       //   y, then z (component c of ray i at [c * ray_stride + i]): a wave that takes
  const double* ray_powers;     //   consecutive rays reads consecutive doubles
  uint64_t ray_stride;          // rays of the launch
  uint64_t first_ray, n_rays, seed;
  uint32_t flags;
  uint32_t chunk;               // rays per hand-out unit of this launch (a multiple of 64, <= ODW_CHUNK)
  unsigned long long* dbg;      // diagnostic builds only (ODW_GRID_STATS): 16 words, or null
  DeviceBatch batch;            // flat kernels' BATCH variants only (n_scenes = 0 otherwise)
  const uint32_t* ray_order;    // mesh and grid kernels: the launch's rays in the order they are handed out (position ->
                                // number of the ray within the launch), or null: by number.  Which wave traces a ray never
                                // shows in its rows (a ray depends on its number only); rays that start alike, traced side
                                // by side, visit the same nodes, facets and cells (odw_capi.hip: presort_rays)
  uint32_t interact_min;        // grid kernel with ray_order: lanes whose walk is over before the wave interacts (<= 1: at once)
  uint32_t refill_min;          // grid kernel: idle lanes that make the wave pop new rays from its ring (0: ODW_GRID_REFILL_MIN)
};

}  // namespace odw
