/*
 * odw_trace.h -- C-ABI of the MI355X-native Monte-Carlo ray-tracing core.
 *
 * The reference (zaphB/freecad.optics_design_workbench) has no FFI: its hot
 * path is the Python method pair
 *     PointSourceProxy._generateRays(mode='true')   point_source.py:659-679
 *     Ray.traceRay(store=...)                        ray.py:36-281
 * driven by GenericSourceProxy.runSimulationIteration (generic_source.py:51)
 * and observed through SimulationResults.addRayHit (results_store.py:641).
 * This header is the boundary a binding for that path would bind: plain
 * pointers and sizes, return codes instead of exceptions.  Each entry point
 * names the reference interface it replaces.
 *
 * All geometry/ray arithmetic is float64 (FreeCAD Vector/Matrix are double).
 * Units: millimetres, radians, wavelength in nm (as in the reference).
 */
#ifndef ODW_TRACE_H
#define ODW_TRACE_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ODW_ABI_VERSION 10

/* ---- return codes ------------------------------------------------------ */
enum {
  ODW_OK = 0,
  ODW_ERR_INVALID = 1,      /* bad argument / inconsistent tables          */
  ODW_ERR_DEVICE = 2,       /* HIP runtime error (see odw_last_error)      */
  ODW_ERR_NO_SCENE = 3,     /* trace before scene/source upload            */
  ODW_ERR_CAPACITY = 4,     /* hit buffer too small for requested fetch    */
  ODW_ERR_UNSUPPORTED = 5,  /* scene feature outside the supported set     */
  ODW_BUSY = 6              /* not an error: the enqueued work has not finished yet (polling calls, v10) */
};

/* ---- primitive kinds (solid primitives; faces are implicit) ------------ */
/* local frames follow FreeCAD's Part primitives:                           */
/*  BOX      [0,L]x[0,W]x[0,H]            params = L, W, H, -               */
/*  SPHERE   centre 0                     params = R, -, -, -               */
/*  CYLINDER axis +z, z in [0,H]          params = R, H, -, -               */
/*  CONE     axis +z, z in [0,H]          params = R1(z=0), R2(z=H), H, -   */
/*  TORUS    axis +z                      params = R1 (ring), R2 (tube)     */
/*  PARABOLOID of revolution, axis +z, vertex at the origin: the solid       */
/*           x^2 + y^2 <= 4 f z, z <= H (a parabolic mirror blank: `README.md`*/
/*           "slotted parabolic mirrors"; FreeCAD builds it as the revolution */
/*           of a parabola)        params = f (focal length), H, 2 sqrt(f H)  */
/*           (= the rim radius, filled in by the library), -                 */
/*  TRIANGLE one facet of a tessellated face (shapes whose surfaces are not */
/*           quadrics: STEP imports, B-splines -- what FreeCAD's            */
/*           `Shape.tessellate(tol)` returns).  No local frame: prim_xform  */
/*           holds v0, v1, v2 in GLOBAL coordinates (9 doubles, 3 unused),  */
/*           counter-clockwise seen from outside the solid; a hit counts if */
/*           it lies within distTol of the facet.  Normals: the facet's, or */
/*           interpolated from tri_normals.  Triangles cannot carry or be   */
/*           referenced by trimming conditions.                             */
enum {
  ODW_PRIM_BOX = 0,
  ODW_PRIM_SPHERE = 1,
  ODW_PRIM_CYLINDER = 2,
  ODW_PRIM_CONE = 3,
  ODW_PRIM_TORUS = 4,
  ODW_PRIM_TRIANGLE = 5,
  ODW_PRIM_PARABOLOID = 6
};

/* face bit positions inside prim_flags >> ODW_FACEMASK_SHIFT               */
/*  BOX: 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z ; CYL/CONE: 0:lateral 1:z=0 2:z=H    */
/*  SPHERE/TORUS/TRIANGLE: 0 ; PARABOLOID: 0:lateral 2:z=H (bit 1 unused)    */
#define ODW_FLAG_FLIP_NORMAL 0x1 /* face normals point INTO the primitive   */
                                 /* (tool of a Part::Cut)                   */
#define ODW_FLAG_CONVEX 0x2      /* the primitive's solid (prim_solid) is convex:*/
                                 /* a box, sphere, cylinder, cone or a Common of */
                                 /* such.  A ray that leaves a convex solid      */
                                 /* (direction . outward normal > 0) cannot meet */
                                 /* it again: its primitives are not tested for  */
                                 /* the next segment.  On ODW_PRIM_TRIANGLE rows:*/
                                 /* the tessellated solid is a convex polyhedron */
                                 /* (closed, facets counter-clockwise from       */
                                 /* outside); the FACET's normal decides the     */
                                 /* exit, not the interpolated one.              */
#define ODW_FLAG_STRICTLY_CONVEX 0x8 /* ODW_PRIM_TRIANGLE rows of an ODW_FLAG_CONVEX solid: no    */
                                 /* edge of the polyhedron bends outward by more than rounding  */
                                 /* (1e-13 of its size), so every point of a facet is on or     */
                                 /* below every other facet's plane.  A ray that travels INSIDE */
                                 /* such a solid is not tested against facets it could only     */
                                 /* meet from outside (results unchanged; a hint for speed).    */
#define ODW_FACEMASK_SHIFT 8

/* ---- optical types: OpticalGroupProxy.OpticalType enumeration order ----- */
/* optical_group.py:32 ['Mirror','Lens','Grating','Absorber','Vacuum']      */
enum {
  ODW_OPT_MIRROR = 0,
  ODW_OPT_LENS = 1,
  ODW_OPT_GRATING = 2,
  ODW_OPT_ABSORBER = 3,
  ODW_OPT_VACUUM = 4
};

#define ODW_MAX_GROUPS 64
#define ODW_MAX_SEQUENCE 100 /* simulation_settings.py:165 range(100)       */

/* Flat SoA scene = what raytracing_cache.py memoises per run (Shape, Shells,
 * Faces, Surface, BoundBox) + the optical group property table
 * (optical_group.py:29-96) + the sequence lists of
 * SimulationSettingsProxy.getTracingSequence (simulation_settings.py:158).   */
typedef struct odw_scene_desc {
  int32_t n_prims;
  const int32_t* prim_type;     /* [n_prims] ODW_PRIM_*                       */
  const int32_t* prim_group;    /* [n_prims] index into group tables          */
  const int32_t* prim_solid;    /* [n_prims] id of the solid (shell) the      */
                                /* primitive bounds, see ODW_FLAG_CONVEX      */
  const int32_t* prim_flags;    /* [n_prims] ODW_FLAG_* | facemask<<8         */
  const double*  prim_xform;    /* [n_prims*12] global->local, rows (R|t)     */
  const double*  prim_params;   /* [n_prims*4]                                */
  const int32_t* prim_cond_off; /* [n_prims+1] CSG trimming conditions        */
  int32_t n_conds;
  const int32_t* cond_prim;     /* [n_conds] primitive the point is tested in */
  const int32_t* cond_inside;   /* [n_conds] 1: must be inside, 0: outside    */

  int32_t n_groups;             /* <= ODW_MAX_GROUPS                          */
  const int32_t* group_type;    /* [n_groups] ODW_OPT_*                       */
  const double*  group_ior;     /* [n_groups] RefractiveIndex                 */
  const double*  group_refl;    /* [n_groups] Reflectivity                    */
  const double*  group_abslen;  /* [n_groups] AbsorptionLength (inf = none)   */
  const int32_t* group_record;  /* [n_groups] RecordHits                      */
  /* grating properties (optical_group.py:82-90); ignored unless GRATING     */
  const int32_t* group_grating_type;   /* [n_groups] 0 reflection 1 transm.  */
  const double*  group_grating_lpm;    /* [n_groups] lines per millimetre    */
  const double*  group_grating_dir;    /* [n_groups*3] GratingLinesOrientation*/
  const int32_t* group_grating_order;  /* [n_groups]                         */

  int32_t seq_enabled;          /* SimulationSettings.SequentialMode          */
  int32_t seq_len;              /* <= ODW_MAX_SEQUENCE                        */
  const uint64_t* seq_mask;     /* [seq_len] bit g set: group g in step       */
  uint64_t ignore_mask;         /* source.IgnoredOpticalElements              */
  /* smooth shading of tessellated faces: unit normals of the surface at the
   * three vertices of each TRIANGLE primitive (rows of other primitives are
   * ignored); NULL: facet normals                                           */
  const double* tri_normals;    /* [n_prims*9] or NULL                        */
  /* which edges of a TRIANGLE primitive are edges of the tessellated FACE (bit 0: edge v0-v2,
   * bit 1: v0-v1, bit 2: v1-v2).  "Within distTol of the face" (ray.py:424-426) widens a facet
   * across those only; across edges it shares with a neighbouring facet of the same face it is
   * closed up to rounding (1e-9 in barycentric units).  NULL: every edge is a face edge.      */
  const int32_t* tri_edges;     /* [n_prims] or NULL                          */
} odw_scene_desc;

/* Point source = PointSourceProxy (point_source.py:32-70) after
 * VectorRandomVariable.compile() in numeric mode
 * (random_number_generator.py:337-464): inverse-CDF tables.                  */
typedef struct odw_source_desc {
  double xform[12];        /* local->global rows (R|t): gpM of _makeRay       */
  double focal_length;     /* finite: (theta,phi) mode; +-inf: (r,phi) mode   */
  double wavelength;       /* nm                                              */
  double power;            /* initial ray power (1)                           */
  int32_t n_phi_knots;     /* odd resolution, e.g. 101                        */
  const double* phi_edges; /* [n_phi_knots]   variableRanges[phi]             */
  const double* phi_cdf;   /* [n_phi_knots]   cumulative sums / last entry    */
  int32_t n_t_knots;       /* e.g. 100001                                     */
  int32_t n_t_rows;        /* n_phi_knots-1, or 1 if density is phi-free      */
  const double* t_edges;   /* [n_t_knots]     variableRanges[theta or r]      */
  const double* t_cdf;     /* [n_t_rows*n_t_knots] each row / its last entry  */
                           /* (random_number_generator.py:441)                */
} odw_source_desc;

/* Stochastic surface of a Mirror / Lens group =
 * OpticalGroupProxy.applyStochasticRayCorrections (optical_group.py:279-323).
 * The reference compiles a VectorRandomVariable over (theta, phi) per hit with
 * the constants theta_in, phi_in = 0, theta_refl, phi_refl = 0
 * (optical_group.py:212-269, 305-307); here the numeric-mode tables
 * (random_number_generator.py:337-464) are tabulated ahead of the launch for
 * a family of n_family equidistant values of ONE constant (family_axis); of
 * the two members around the hit's constant, member k0 + 1 is taken with
 * probability = the fractional position between the knots (a uniform of its
 * own: counter word 3 = 17 + kind), and sampled exactly like a source.
 *   kind PRIMARY: Reflected- (mirror) / RefractedProbabilityDensity (lens):
 *     out = Rot(normal, phi) * Rot(normal x dirIn, theta) * normal
 *   kind MODIFY : RayModificationProbabilityDensity, applied afterwards:
 *     out = Rot(out, phi) * Rot(out x dirIn, theta) * out
 * (Rot(axis, angle) = FreeCAD Rotation; a zero axis is the identity.)
 * Uniforms: Philox4x32-10, key = seed, counter = (ray_lo, ray_hi,
 * intersection ordinal 1.., 1 + kind); words (0,1) -> u_phi, (2,3) -> u_theta.
 * mu: several samplers may serve one (group, ODW_SURF_PRIMARY) of a LENS whose density names both theta_in
 *   and theta_refl -- the two are tied by Snell's law, theta_refl = asin(mu sin theta_in), mu = n1 / n2 of the
 *   hit (ray.py:171-199) --: one family over theta_in per value of mu; the hit takes the sampler whose mu is
 *   its own (nearest in log mu); mu = -1 serves total reflection (theta_refl = pi - theta_in), mu = 0 every hit.
 * Atoms: DiracDelta terms of the density (the reference's analytic mode with discrete events,
 *   random_number_generator.py:204-320): with probability atom_mass[member][j] the draw is
 *   theta = a0 + a1 theta_in + a2 theta_refl, phi = b (rule 0) or uniform over [lo, hi] (rule 1); else the
 *   tables.  Decided by u_phi: atom j owns [sum of the masses before it, + its own); what is left of the
 *   unit interval is stretched back to [0, 1) and goes on as u_phi of the tables (or as the uniform of rule 1). */
enum { ODW_SURF_PRIMARY = 0, ODW_SURF_MODIFY = 1 };
enum { ODW_SURF_AXIS_NONE = 0, ODW_SURF_AXIS_THETA_IN = 1, ODW_SURF_AXIS_THETA_REFL = 2 };
#define ODW_SURF_MAX_ATOMS 4
typedef struct odw_surface_sampler_desc {
  int32_t group;           /* index into the group tables (Mirror or Lens)    */
  int32_t kind;            /* ODW_SURF_PRIMARY / ODW_SURF_MODIFY              */
  int32_t family_axis;     /* ODW_SURF_AXIS_*; NONE: n_family must be 1       */
  int32_t n_family;        /* member k is compiled for the constant value     */
  double family_lo;        /*   family_lo + k (family_hi-family_lo)/(n_family-1) */
  double family_hi;
  int32_t n_phi_knots;
  const double* phi_edges; /* [n_phi_knots]                                   */
  const double* phi_cdf;   /* [n_family*n_phi_knots] each / its last entry    */
  int32_t n_t_knots;
  int32_t n_t_rows;        /* n_phi_knots-1, or 1 if the density is phi-free  */
  const double* t_edges;   /* [n_t_knots]                                     */
  const double* t_cdf;     /* [n_family*n_t_rows*n_t_knots]                   */
  double mu;               /* > 0: n1 / n2 this sampler is for; -1: total     */
                           /* reflection on a lens; 0: every hit              */
  int32_t n_atoms;         /* <= ODW_SURF_MAX_ATOMS discrete events, or 0     */
  const double* atom_mass; /* [n_family*n_atoms] probability per member       */
  const double* atom_theta;/* [n_atoms*3] a0, a1 (x theta_in), a2 (x theta_refl) */
  const double* atom_phi;  /* [n_atoms*3] rule (0 fixed, 1 uniform), lo, hi   */
} odw_surface_sampler_desc;

/* Surface source = SurfaceSourceProxy._generateRays(mode='true')
 * (surface_source.py:519-553): a ray starts at a uniformly distributed point
 * of the emitting faces (the reference picks a face by area and draws (u,v)
 * from a sampled area-element grid, :440-464, :394-412; here the analytic
 * faces are sampled exactly), its direction is
 *   Rot(normal, phi) * Rot(tangent, theta) * normal        (:104-106)
 * with theta from the scalar random variable of PowerDensity over ThetaDomain
 * (numeric mode table) and phi uniform in [0, 2 pi).
 * Emitting solids are primitives in the scene's encoding with their own
 * boolean trimming conditions; a sampled point that a condition trims away
 * is rejected together with its face choice, so faces weigh in by their
 * TRIMMED area.  Uniforms: Philox4x32-10, key = seed, counter =
 * (ray_lo, ray_hi, attempt, 3) -> face, acceptance; (.., attempt, 4) -> the
 * two face coordinates; (ray_lo, ray_hi, 0, 5) -> theta, phi.
 * Faces of tessellated shapes (BRep imports) emit facet by facet: an
 * ODW_PRIM_TRIANGLE primitive (prim_xform = v0, v1, v2 in global coordinates
 * as in odw_scene_desc, no conditions) with one face of the facet's area; the
 * point is v0 + a (v1-v0) + b (v2-v0) with (a, b) = the two face coordinates
 * folded into a + b <= 1, the normal the facet's or, with tri_normals, the
 * normalised barycentric mix of the three vertex normals, the tangent the
 * part of v1-v0 perpendicular to it.                                        */
typedef struct odw_surface_source_desc {
  double wavelength;            /* nm                                         */
  double power;                 /* initial ray power (1)                      */
  double dist_tol;              /* SurfaceSourceProxy._getDistTol, :111-116   */
  int32_t n_prims;
  const int32_t* prim_type;     /* [n_prims] ODW_PRIM_*                       */
  const int32_t* prim_flags;    /* [n_prims] ODW_FLAG_FLIP_NORMAL             */
  const double*  prim_xform;    /* [n_prims*12] global->local rows (R|t)      */
  const double*  prim_params;   /* [n_prims*4]                                */
  const int32_t* prim_cond_off; /* [n_prims+1]                                */
  int32_t n_conds;
  const int32_t* cond_prim;     /* [n_conds] index into THIS primitive list   */
  const int32_t* cond_inside;   /* [n_conds]                                  */
  int32_t n_faces;
  const int32_t* face_prim;     /* [n_faces]                                  */
  const int32_t* face_id;       /* [n_faces] face bit position (see above)    */
  const double*  face_area;     /* [n_faces] area of the UNTRIMMED face       */
  int32_t n_t_knots;
  const double* t_edges;        /* [n_t_knots] theta                          */
  const double* t_cdf;          /* [n_t_knots] / last entry                   */
  const double* tri_normals;    /* [n_prims*9] unit vertex normals of TRIANGLE
                                 * primitives (other rows ignored), or NULL   */
} odw_surface_source_desc;

/* Ray.traceRay keyword arguments + settings (ray.py:36-73, 283-288).        */
typedef struct odw_limits {
  double max_ray_length;    /* MaxRayLengthScale * settings.MaxRayLength      */
  int32_t max_intersections;/* MaxIntersectionsScale * MaxIntersections       */
  double dist_tol;          /* max(DistanceTolerance, 1e-6)                   */
  double power_tol;         /* 1e-6                                           */
} odw_limits;

/* Up-front detector binning (cartesian): the reference bins post hoc
 * (hits.py:176-193 -> histogram.py:24-57); here the plane is fixed before
 * tracing.  Hit p on a RecordHits group g is binned at
 *   x = (p-origin).ex, y = (p-origin).ey,
 *   ix = floor((x-x_lo)*nx/(x_hi-x_lo)), same for y; outside -> overflow
 * counter.  group < 0: all recording groups.                                 */
typedef struct odw_detector_desc {
  int32_t group;
  double origin[3], ex[3], ey[3];
  double x_lo, x_hi, y_lo, y_hi;
  int32_t nx, ny;
} odw_detector_desc;

/* One recorded hit: the row layout SimulationResults.flush pickles
 * (results_store.py:405-457): points, directions, powers, isEntering.
 * 64 bytes.  tag = ray_index | group<<48 | isEntering<<63.                   */
typedef struct odw_hit {
  double point[3];
  double direction[3]; /* incoming direction (pre-interaction), ray.py:132   */
  double power;        /* after medium absorption, before this surface       */
  uint64_t tag;
} odw_hit;

#define ODW_HIT_RAY(tag)      ((tag) & 0xFFFFFFFFFFFFull)
#define ODW_HIT_GROUP(tag)    (((tag) >> 48) & 0x7FFF)
#define ODW_HIT_ENTERING(tag) ((tag) >> 63)

/* One ray segment as Ray.traceRay yields it (ray.py:104-117) and
 * SimulationResultsSingleRay.addSegment keeps it for sources with RecordRays
 * (generic_source.py:78-118, results_store.py:232-257): (p1, p2), the power
 * at p1 and the medium the segment runs through.  A ray that finds no
 * intersection ends with p2 = p1 + direction * max_ray_length (ray.py:107).
 * 64 bytes.  tag = ray_index (40 bits) | ordinal << 40 (12 bits; 0 = first
 * segment) | (medium group + 1) << 52 (0 = vacuum).                         */
typedef struct odw_segment {
  double p1[3];
  double p2[3];
  double power;
  uint64_t tag;
} odw_segment;

#define ODW_SEG_RAY(tag)     ((tag) & 0xFFFFFFFFFFull)
#define ODW_SEG_ORDINAL(tag) (((tag) >> 40) & 0xFFF)
#define ODW_SEG_MEDIUM(tag)  ((int32_t)(((tag) >> 52) & 0xFFF) - 1)
#define ODW_SEG_MAX_RAY      (1ull << 40)   /* first_ray + n_rays of a recording launch */
#define ODW_SEG_MAX_ORDINAL  4096           /* max_intersections of a recording launch  */

/* counters (u64), SimulationResults counters results_store.py:306-310       */
enum {
  ODW_CNT_TRACED_RAYS = 0,   /* incrementRayCount, generic_source.py:141     */
  ODW_CNT_RECORDED_HITS = 1, /* addRayHit on RecordHits groups               */
  ODW_CNT_SEGMENTS = 2,      /* nearest-hit queries                          */
  ODW_CNT_ESCAPED = 3,       /* no intersection found, ray.py:105            */
  ODW_CNT_DIED = 4,          /* power < powerTol, ray.py:280                 */
  ODW_CNT_CAPPED = 5,        /* numIntersections >= max, ray.py:96           */
  ODW_CNT_HIST_OVERFLOW = 6, /* recorded hits outside the detector window    */
  ODW_CNT_HITS_DROPPED = 7,  /* hit list capacity exceeded                   */
  ODW_CNT_GRATING_IN_MEDIUM = 8, /* rays that entered a transmission grating
                              * inside a medium: the reference raises
                              * ValueError there (ray.py:234-237); the ray is
                              * ended (also counted under DIED), the caller
                              * turns a non-zero count into that exception    */
  ODW_CNT_COUNT = 9
};

/* trace flags */
#define ODW_TRACE_RECORD_HITS 0x1 /* append odw_hit rows                      */
#define ODW_TRACE_HISTOGRAM   0x2 /* bin into the detector histogram          */
#define ODW_TRACE_RECORD_SEGMENTS 0x4 /* append odw_segment rows (RecordRays) */

typedef struct odw_ctx odw_ctx;

/* lifecycle ------------------------------------------------------------- */
int odw_abi_version(void);
int odw_create(int device, odw_ctx** out);
void odw_destroy(odw_ctx* ctx);
const char* odw_last_error(const odw_ctx* ctx); /* NULL ctx: global message */

/* scene bake upload: replaces raytracing_cache.py:43-114 + find.py:79-104  */
int odw_upload_scene(odw_ctx* ctx, const odw_scene_desc* scene);
/* replaces PointSourceProxy._getVrv/_rvArgs result, point_source.py:277-386 */
int odw_upload_source(odw_ctx* ctx, const odw_source_desc* source);
/* replaces SurfaceSourceProxy._generateRays(mode='true'), surface_source.py:519-553;
 * the most recently uploaded source (point or surface) feeds odw_trace      */
int odw_upload_surface_source(odw_ctx* ctx, const odw_surface_source_desc* source);
/* replaces OpticalGroupProxy._getVrv + per-hit compile, optical_group.py:212-323;
 * call after odw_upload_scene (which clears them); n = 0 clears             */
int odw_upload_surface_samplers(odw_ctx* ctx, const odw_surface_sampler_desc* samplers,
                                int32_t n);
/* Philox key of the surface draws in odw_trace_rays launches (odw_trace uses
 * its own seed argument); default 0                                         */
int odw_set_surface_seed(odw_ctx* ctx, uint64_t seed);
/* wavelength (nm) of the following launches: Ray(..., wavelength=...) of
 * ReplaySourceProxy._generateRays (replay_source.py:155); odw_upload_source
 * resets it to the source's own                                              */
int odw_set_wavelength(odw_ctx* ctx, double wavelength_nm);
int odw_set_limits(odw_ctx* ctx, const odw_limits* limits);
int odw_set_detector(odw_ctx* ctx, const odw_detector_desc* det);
/* capacity of the device hit list in rows (0 frees it)                     */
int odw_reserve_hits(odw_ctx* ctx, uint64_t capacity);
/* capacity of the device segment list in rows (0 frees it); needed by
 * launches with ODW_TRACE_RECORD_SEGMENTS                                   */
int odw_reserve_segments(odw_ctx* ctx, uint64_t capacity);

/* replaces one or many runSimulationIteration calls (generic_source.py:51):
 * rays first_ray .. first_ray+n_rays-1 of the global Philox stream `seed`
 * are generated (point_source.py:659-679, random_number_generator.py:467),
 * traced (ray.py:36-281) and recorded.  Asynchronous on the context stream. */
int odw_trace(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, uint64_t seed,
              uint32_t flags);
/* trace explicit initial conditions instead of the sampler ("useInitial-
 * Conditions", generic_source.py:59; fan mode rays): origin/dir [n*3].     */
int odw_trace_rays(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays,
                   const double* origins, const double* directions,
                   const double* powers, uint32_t flags);
int odw_sync(odw_ctx* ctx);

/* results --------------------------------------------------------------- */
int odw_reset_results(odw_ctx* ctx); /* zero counters, hits, segments, histogram */
int odw_reset_hits(odw_ctx* ctx);    /* recycle the hit list only (flush)   */
int odw_fetch_counters(odw_ctx* ctx, uint64_t* out, int32_t n);
int odw_hit_count(odw_ctx* ctx, uint64_t* n);
/* copies min(n_hits, capacity) rows, sorted by (ray index, bounce order)   */
int odw_fetch_hits(odw_ctx* ctx, odw_hit* out, uint64_t capacity, uint64_t* n);
int odw_fetch_histogram(odw_ctx* ctx, uint64_t* out, uint64_t n_bins);
/* Overlapped row fetch for continuous runs that keep every hit (the reference
 * buffers hits while the workers trace on and flushes them every few seconds,
 * results_store.py:405-457): two hit lists.  odw_swap_hit_lists puts the list
 * traced into so far aside and makes the other one current (allocated with
 * the same room on first use; the caller recycles it with odw_reset_hits);
 * odw_fetch_swapped_hits copies the rows of the list put aside to the host on
 * a copy stream of its own -- it waits for the launches issued before the
 * swap only, so a launch issued after it runs while the rows cross PCIe.
 * Rows arrive in append order (unordered across rays; a ray's rows in bounce
 * order); once lists are swapped appends are dense (no reserved blocks).    */
int odw_swap_hit_lists(odw_ctx* ctx);
int odw_fetch_swapped_hits(odw_ctx* ctx, odw_hit* out, uint64_t capacity, uint64_t* n);
/* end of a streamed run: frees the list put aside; later launches reserve
 * hit-list blocks per wave again (the current list and its rows stay)        */
int odw_release_swapped_hits(odw_ctx* ctx);
/* page-locked host memory for the destination of row fetches (the copy engine
 * writes it directly: no staging through the runtime's own pinned buffers)  */
/* v10: free and total memory of the context's device (hipMemGetInfo): what a
 * sweep sizes its groups by and a run decides by whether its rows may stay in
 * HBM -- instead of a fixed budget that ignores other contexts and ranks on
 * the same GPU.                                                             */
int odw_mem_info(odw_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
int odw_host_alloc(odw_ctx* ctx, uint64_t bytes, void** out);
/* ctx may be NULL: arrays handed to the caller (hit columns of a run kept in memory) outlive their context */
int odw_host_free(odw_ctx* ctx, void* p);
/* rows in the segment list and rows that did not fit into it               */
int odw_segment_count(odw_ctx* ctx, uint64_t* n, uint64_t* dropped);
/* copies the rows sorted by (ray index, ordinal); NULL out: count only      */
int odw_fetch_segments(odw_ctx* ctx, odw_segment* out, uint64_t capacity, uint64_t* n);
int odw_reset_segments(odw_ctx* ctx); /* recycle the segment list            */
/* sampler only (diagnostics/tests): theta-or-radius and phi of each ray    */
int odw_sample(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, uint64_t seed,
               double* theta_out, double* phi_out);

/* initial conditions only ("returnInitialConditions", generic_source.py:57):
 * origin and direction [n*3] of each ray of the uploaded source             */
int odw_generate_rays(odw_ctx* ctx, uint64_t first_ray, uint64_t n_rays, uint64_t seed,
                      double* origins, double* directions);

/* post-hoc detector binning on the rows in HBM --------------------------------
 * Replaces the array work of Hits.histogram (jupyter_utils/hits.py:176-193):
 * detectPlaneNormal's thinned sample (:96-113), planeProject3dPoints (:62-94),
 * Histogram.__init__ (jupyter_utils/histogram.py:24-57: median origin,
 * numpy.histogram2d of (X, Y) or of (arctan2(X, Y), hypot(X, Y))).  The plane
 * search itself works on <= 300 rows and stays with the caller.  Call order:
 * select -> gather (sample) -> project -> [range] -> bin; any launch, reset or
 * fetch of the hit list ends the selection.                                  */
/* put rows recorded earlier (a run folder's `*-hits.pkl` files,
 * RawFolder.loadHits, freecad_document.py:1485-1504) back into the device hit
 * list, replacing its content, so that they can be binned there             */
int odw_load_hits(odw_ctx* ctx, const odw_hit* rows, uint64_t n);
/* the rows of one recording group (-1: all) in (ray index, bounce) order;
 * n_leaving: how many of them have isEntering == 0 (hits.py:108-111)        */
int odw_hits_select(odw_ctx* ctx, int32_t group, uint64_t* n_rows, uint64_t* n_leaving);
/* rows 0, stride, 2 stride, ... of the selection (entering_only: of its rows
 * with isEntering != 0) = numpy's a[::stride]; NULL out: count only         */
int odw_hits_gather(odw_ctx* ctx, int32_t entering_only, uint64_t stride, odw_hit* out,
                    uint64_t capacity, uint64_t* n);
/* the selection as the arrays the reference pickles per (source, object)
 * (results_store.py:405-457: points, directions, powers, isEntering; the ray
 * index for per-ray metadata) -- gathered on the device, in (ray, bounce)
 * order; any pointer may be NULL (that column is skipped); *n = rows of the
 * selection                                                                 */
int odw_hits_columns(odw_ctx* ctx, double* points, double* directions, double* powers, int64_t* is_entering,
                     int64_t* ray_index, uint64_t capacity, uint64_t* n);
/* X = p . ex, Y = p . ey for every selected row (key 0: points, 1: directions;
 * ex, ey: unit vectors, [3] each); stats[8] = for X then Y: the two middle
 * values of the sorted column (numpy.median = their mean), minimum, maximum */
int odw_hits_project(odw_ctx* ctx, int32_t key, const double* ex, const double* ey, double* stats);
/* range[4] = min, max of the first and of the second binned coordinate:
 * cartesian (X - origin[0], Y - origin[1]); polar (arctan2 of those two,
 * their hypot) -- what numpy takes as outer edges for integer bin counts    */
int odw_hits_range(odw_ctx* ctx, int32_t polar, const double* origin, double* range);
/* counts[(n_a-1)*(n_b-1)] (first coordinate major) with numpy.histogramdd's
 * rule: bin = searchsorted(edges, v, 'right') - 1, the last edge closed      */
int odw_hits_bin(odw_ctx* ctx, int32_t polar, const double* origin, const double* edges_a, int32_t n_a,
                 const double* edges_b, int32_t n_b, uint64_t* counts);
/* mean[3] and variance[3] (about the mean) of the selected rows' points     */
int odw_hits_moments(odw_ctx* ctx, double* mean, double* var);
/* The screen of the plane search of `Hits.detectPlaneNormal` (jupyter_utils/hits.py:108-137: the direction along
 * which the thinned cloud has the smallest extent -- a 30 x 30 grid over half the unit sphere, then 10 x 10 grids
 * around the best candidate): for cloud[n][3] and one grid, extent[i * n_phi + j] = max - min of point . normal with
 * normal = (cos phi_j sin theta_i, sin phi_j sin theta_i, cos theta_i), the reference's candidate order.  The
 * reference takes the first minimum of extents summed in numpy's order; the caller therefore evaluates the
 * candidates within rounding of the smallest screened extent again with numpy and decides there.  No context, no
 * device work; thread-safe (the measuring threads of a parameter sweep call it without the interpreter lock).   */
int odw_plane_screen(const double* cloud, uint64_t n, const double* phis, int32_t n_phi, const double* thetas, int32_t n_theta,
                     double* extent);
/* the same for n_clouds clouds at once, each with its own grid of the same size (phis [n_clouds][n_phi], thetas
 * [n_clouds][n_theta], extent [n_clouds][n_theta * n_phi]), on a few threads of the library: the plane searches of the
 * scenes of a batch launch advance level by level, one call per level (v9)                                        */
int odw_plane_screen_batch(const double* const* clouds, const uint64_t* n, int32_t n_clouds, const double* phis, int32_t n_phi,
                           const double* thetas, int32_t n_theta, double* extent);

/* scene-compiled kernels -------------------------------------------------------
 * The reference prepares a scene once per run and reuses it for every ray
 * (simulation/raytracing_cache.py:92-111 cachedShape / cachedFaces /
 * cachedBoundBox ..., cacheClear :36).  Here, for a scene of <= 64 analytic
 * primitives the library can compile the ray loop against the scene itself
 * (hiprtc: 0.5 - 2 s up to 16 primitives, 17 s for 61; cached per process and
 * on disk): primitive loop unrolled,
 * type dispatch / face masks / trimming lists / optical types folded.
 * Float64 values (frames, parameters, boxes, optical constants) are still read
 * from the uploaded tables: one kernel per scene STRUCTURE, parameter sweeps
 * reuse it.
 * The mode is sticky: it applies to the uploaded scene (bound at once if scene
 * and limits are there, else at the next launch) and to every scene uploaded
 * later.  Scenes outside the domain (facets, > 64 primitives) and launches
 * that record segment rows keep the generic kernels --
 * that is not an error.  Results are those of the generic kernel, bit for bit.
 * If the kernel cannot be built (hiprtc missing, a compiler error) the call
 * returns ODW_ERR_DEVICE with the reason in odw_last_error; launches go on
 * with the generic kernels.
 * ODW_KERNEL_CACHE: directory of the disk cache (default ~/.cache/odw_trace,
 * empty string: none).                                                       */
#define ODW_COMPILE_OFF 0
#define ODW_COMPILE_STRUCTURE 1   /* compile when the scene is bound (the call / the next launch waits)      */
#define ODW_COMPILE_AUTO 2        /* never wait: a kernel from a cache is bound at once; otherwise the scene
                                   * is traced by the generic kernels, a thread compiles once 5e7 rays
                                   * (ODW_SPEC_HOT_RAYS) were traced with it, and the launch after it has
                                   * finished takes the compiled kernel -- same rows bit for bit           */
int odw_compile_scene(odw_ctx* ctx, int32_t mode);
/* bound: the mode of the kernel the next eligible launch runs (0: generic);
 * compile_seconds: of the bound kernel (0 if it came from a cache);
 * cache_hit: 0 compiled now, 1 process cache, 2 disk cache                   */
int odw_compiled_info(odw_ctx* ctx, int32_t* bound, double* compile_seconds, int32_t* cache_hit);
/* no device needed: writes the scene's header (NUL-terminated, truncated to
 * header_capacity) and compiles the kernel for `arch` (NULL: "gfx950");
 * code_bytes = size of the code object.  ODW_ERR_UNSUPPORTED: the scene is
 * outside the flat kernel's domain.                                          */
int odw_compile_check(const odw_scene_desc* scene, const odw_limits* limits, int32_t mode, const char* arch,
                      char* header_out, uint64_t header_capacity, uint64_t* code_bytes);
/* v10: what odw_upload_scene + the first launch prepare -- validation, host
 * tables, boxes, the choice among flat loop / grid / binary tree / eight-wide
 * tree and their construction -- WITHOUT a device (the tables go to host memory).
 * Replaces nothing of the reference (its acceleration structure is OCC's);
 * exists so that these builders run under a CPU sanitizer and so that a caller
 * can ask what a scene will be traced with.  structure: 0 flat, 1 grid, 2
 * binary tree, 3 eight-wide tree; sizes [6]: primitives, tree nodes, grid
 * cells, grid items, LDS bytes of a grid block, dead primitives.            */
int odw_build_check(const odw_scene_desc* scene, const odw_limits* limits, int32_t* structure, uint64_t* sizes);

/* ---- batches: many scenes of ONE structure in one launch (v9) -----------
 * Replaces the loop of a parameter sweep (examples/1-getting-started/
 * optimize-spotsize.ipynb cell 9; jupyter_utils/parameter_sweeper.py): one
 * `runSimulation` per parameter value there, one launch for all values here.
 * The scenes must agree in everything but their numbers: the same primitives,
 * trimming lists, groups, optical types, sequence (ODW_ERR_UNSUPPORTED names
 * the difference).  Scene 0 becomes the context's scene; limits first (the
 * boxes carry the tolerance), flat kernels only (analytic scenes of up to 64
 * primitives, no stochastic surfaces).
 * odw_trace_batch traces rays first_ray ... first_ray + rays_per_scene - 1 in
 * EVERY scene (the rows of a scene are those of odw_trace on that scene
 * alone, tags included -- the scenes share their rays, so from three scenes
 * on their initial conditions are generated once per launch and read by
 * every scene, instead of once per scene); a scene's rows go to its own segment of the batch's
 * hit list (room for rows_per_scene rows each); counters add up over the
 * scenes; no detector histogram.  odw_batch_select makes a segment the
 * context's hit list for odw_hit_count / odw_fetch_hits / odw_hits_* (scene
 * < 0, or any other trace / reserve / reset call: back to the context's own
 * list).  odw_batch_rows: rows recorded per scene, and (optional) the slots
 * asked for -- above the segment's room rows were dropped (counter
 * ODW_CNT_HITS_DROPPED): trace again with more room.  The segments belong to
 * the LAST odw_trace_batch: after one that failed, or one without
 * ODW_TRACE_RECORD_HITS, odw_batch_select / odw_batch_rows / odw_batch_hits_*
 * answer ODW_ERR_INVALID instead of handing out an older launch's rows.      */
int odw_upload_scene_batch(odw_ctx* ctx, const odw_scene_desc* scenes, int32_t n_scenes);
int odw_trace_batch(odw_ctx* ctx, uint64_t first_ray, uint64_t rays_per_scene, uint64_t seed, uint32_t flags,
                    uint64_t rows_per_scene);
/* v10: room for batches of up to n_scenes x rays_per_scene rays x rows_per_scene
 * rows (and for the post-hoc chain below) allocated now instead of when a
 * larger batch first arrives: growing a buffer in the middle of a sweep waits
 * for every stream of the device.                                           */
int odw_batch_reserve(odw_ctx* ctx, int32_t n_scenes, uint64_t rays_per_scene, uint64_t rows_per_scene);
int odw_batch_select(odw_ctx* ctx, int32_t scene);
int odw_batch_rows(odw_ctx* ctx, uint64_t* rows, uint64_t* wanted, int32_t n);
/* The post-hoc steps (odw_hits_select / gather / project / bin / moments:
 * Hits.histogram, jupyter_utils/hits.py:96-193, histogram.py:24-57) for ALL
 * segments of the last odw_trace_batch at once: the same kernels per segment,
 * ONE synchronisation per step for the whole batch; arrays are indexed by
 * scene.  ordered[s] = 0: scene s needs the per-segment calls (a ray with two
 * selected rows, or a sample that needs the entering rows of a mixed list).
 * sample: rows [::strides[s]] if strides are given, else
 * points[::1 + n / limit] (their directions are the sample's
 * directions: every row enters, or leaving rows are the majority), [S][cap].
 * project: key = points; stats [S][8] as odw_hits_project, moments [S][6] =
 * mean (3), variance (3) as odw_hits_moments; skip[s] != 0 leaves a scene out.
 * bin: the same edges for all scenes, each about its own origin [S][2];
 * counts [S][(n_a - 1) (n_b - 1)].                                           */
int odw_batch_hits_select(odw_ctx* ctx, int32_t group, uint64_t* n_rows, uint64_t* n_leaving, int32_t* ordered);
int odw_batch_hits_sample(odw_ctx* ctx, uint64_t limit, const uint64_t* strides, odw_hit* rows, uint64_t cap, uint64_t* n_out);
int odw_batch_hits_project(odw_ctx* ctx, const double* ex, const double* ey, const int32_t* skip, double* stats, double* moments);
int odw_batch_hits_bin(odw_ctx* ctx, int32_t polar, const double* origins, const double* edges_a, int32_t n_a,
                       const double* edges_b, int32_t n_b, uint64_t* counts);
/* v10: the same chain as TWO stream-ordered pieces that are enqueued and
 * polled instead of waited for step by step -- every decision between two
 * kernels (row counts, the histogram bins that hold the middle ranks, the
 * ranks among the median candidates, the origin) is taken on the device, the
 * only hand-over to the host is the plane search of Hits.detectPlaneNormal
 * (jupyter_utils/hits.py:96-174) on the thinned sample.  A parameter sweep
 * (optimize-spotsize.ipynb cells 8 - 11) keeps several contexts' chains in
 * flight from one host thread.
 *   begin     select + the sample points[::1 + n / limit] of every ordered
 *             scene (limit <= 4096), enqueued behind the batch launch
 *   sampled   wait != 0: waits; else ODW_BUSY while the piece is under way.
 *             n_rows / n_leaving / ordered / n_sample: [S]; rows: [S][cap],
 *             cap >= limit + 8
 *   measure   planes (ex, ey: [S][3]; skip[s] != 0 leaves a scene out) ->
 *             projection, medians, moments, histogram of (polar ? (arctan2(X,
 *             Y), hypot) : (X, Y)) about the median origin, enqueued
 *   measured  stats [S][8], moments [S][6], origins [S][2], counts
 *             [S][(n_a - 1) (n_b - 1)], flags [S] (non-zero: this scene needs
 *             the per-segment calls -- more median candidates than the device
 *             ranks, i.e. a cloud piled up on one value); keep != 0: the rows
 *             [::max(1, n / keep)] of every ordered scene ride along
 *             (keep_rows [S][keep_cap], keep_cap >= 2 keep + 8; n_keep [S]):
 *             the sample a notebook that traces `keep` rays per value sees  */
int odw_batch_hits_begin(odw_ctx* ctx, int32_t group, uint64_t limit);
int odw_batch_hits_sampled(odw_ctx* ctx, int32_t wait, uint64_t* n_rows, uint64_t* n_leaving, int32_t* ordered, odw_hit* rows,
                           uint64_t cap, uint64_t* n_sample);
int odw_batch_hits_measure(odw_ctx* ctx, const double* ex, const double* ey, const int32_t* skip, int32_t polar,
                           const double* edges_a, int32_t n_a, const double* edges_b, int32_t n_b, uint64_t keep);
int odw_batch_hits_measured(odw_ctx* ctx, int32_t wait, double* stats, double* moments, double* origins, uint64_t* counts,
                            uint32_t* flags, odw_hit* keep_rows, uint64_t keep_cap, uint64_t* n_keep);

/* ---- a run's rows kept in HBM (v9) ---------------------------------------
 * The reference keeps a run's hits in its run folder and loads them all into
 * host arrays to analyse them (freecad_document.py:1485-1504 -> Hits); here a
 * run can keep its rows on the device: odw_archive_append adds the rows of
 * src's hit list (src may be ctx itself, or another context of the same device
 * whose launch has just finished) to ctx's archive, device to device;
 * odw_archive_select(ctx, 1) makes the archive ctx's hit list for
 * odw_hit_count / odw_fetch_hits / odw_hits_* (0, or any trace / reserve /
 * reset call: back to the context's own list); odw_archive_reset drops it.   */
int odw_archive_append(odw_ctx* ctx, odw_ctx* src, uint64_t* total_rows);
int odw_archive_select(odw_ctx* ctx, int32_t on);
int odw_archive_reset(odw_ctx* ctx);

/* device-side handles for collectives (RCCL reduce through torch).  The
 * counters and the detector histogram live in ONE block of 64-bit words,
 * [hist_offset_words words: the ODW_CNT_* counters, zero-padded][n_bins bins],
 * so that a multi-GPU job sums everything its ranks produced with a single
 * reduce (v9; replaces the file-system merge of the reference's workers,
 * simulation_loop.py:450-507, freecad_document.py:1491-1504).  The block moves
 * when odw_set_detector asks for more bins: take the pointer after it.      */
int odw_device_results(odw_ctx* ctx, void** dptr, uint64_t* n_words, uint64_t* hist_offset_words);
/* the two parts of that block by themselves                                 */
int odw_device_histogram(odw_ctx* ctx, void** dptr, uint64_t* n_bins);
int odw_device_counters(odw_ctx* ctx, void** dptr, uint64_t* n);
int odw_stream(odw_ctx* ctx, void** hip_stream);

/* timing of the trace kernel with HIP events on the context stream         */
int odw_timing_enable(odw_ctx* ctx, int on);
int odw_timing_read(odw_ctx* ctx, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ODW_TRACE_H */
