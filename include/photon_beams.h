/*
 * photon_beams.h — C ABI of the MI355X photon-beam radiance path.
 *
 * The reference (RaeJ/Raytracer) has no FFI: its boundary is the set of free
 * functions that Source/skeleton.cpp calls on shared globals.  Each entry point
 * below names the reference interface it replaces (file:line under the
 * reference tree).  All functions return PB_OK (0) or a negative pb_status and
 * never throw, print or exit.  Pointers are caller-owned HOST memory unless the
 * parameter name starts with d_ (then it is device memory on the context's GPU).
 * A context belongs to one GPU and one host thread at a time.
 *
 * Coordinates are the reference's camera space (world space multiplied by the
 * matrix of TransformationMatrix, skeleton.cpp:512-532); every camera ray
 * starts at (0,0,0) as in DrawStandard (skeleton.cpp:346).
 */
#ifndef PHOTON_BEAMS_H
#define PHOTON_BEAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pb_status {
  PB_OK = 0,
  PB_ERR_ARG = -1,        /* null pointer, bad size, bad mode                    */
  PB_ERR_STATE = -2,      /* call order: scene/segments/LBVH not set yet         */
  PB_ERR_DEVICE = -3,     /* a HIP call failed; see pb_last_error()              */
  PB_ERR_NO_DEVICE = -4,  /* no usable GPU: there is no CPU fallback             */
  PB_ERR_CAPACITY = -5    /* a fixed-size buffer (output array, queue) is small  */
} pb_status;

/* Run-time form of the compile-time switches and globals of Source/Control.h:4-50
 * (same names, lower case; same defaults from pb_control_defaults). */
typedef struct pb_control {
  int32_t screen_width;     /* SCREEN_WIDTH   512 */
  int32_t screen_height;    /* SCREEN_HEIGHT  512 */
  int32_t ssaa;             /* SSAA           1   */
  int32_t photon_number;    /* PHOTON_NUMBER  1000 */
  int32_t bounces;          /* BOUNCES        0   */
  int32_t adaptive;         /* ADAPTIVE       false */
  int32_t short_beams;      /* SHORT_BEAMS    false */
  int32_t short_view;       /* SHORT_VIEW     false */
  int32_t scatter;          /* SCATTER        true  */
  int32_t one_dimensional;  /* ONE_DIMENSIONAL false */
  int32_t fixed_radius;     /* FIXED_RADIUS   false */
  int32_t heterogeneous;    /* HETEROGENEOUS  false */
  float   radius;           /* RADIUS         0.01f */
  float   absorbed;         /* ABSORBED       0.0f  */
  uint32_t seed;            /* SEED           42    */
  float   light_position[4];/* (0,-0.8,-0.5,1)      */
  float   light_power[3];   /* 0.5*(1,1,1)          */
  float   absorption_c;     /* 0.035                */
  float   scattering_c;     /* 0.005; extinction_c = absorption_c + scattering_c */
  float   camera[4];        /* (0,0,-3,1)           */
  float   theta[3];         /* skeleton.cpp:25 (0,0,0) */
} pb_control;

/* One photon-beam segment: the fields of PhotonSeg (Intersections.h:36-52) that
 * the radiance estimate reads, plus the box that gates it and its beam's energy.
 * 28 little-endian words = 112 bytes; also the on-disk dump format. */
typedef struct pb_segment {
  float start[3];       /* PhotonSeg.start.xyz                                   */
  float end[3];         /* PhotonSeg.end.xyz                                     */
  float orig_start[3];  /* PhotonSeg.orig_start.xyz (start of the whole beam)    */
  float radius;         /* PhotonSeg.radius                                      */
  float cull_lo[3];     /* box whose entry distance gates the segment, as the    */
  float cull_hi[3];     /* min_b/max_b of HitBoundingBox (Intersections.h:85-86):*/
                        /* the leaf-node box of the reference tree, or the       */
                        /* segment's own PhotonSeg.min/max box                   */
  float energy[3];      /* PhotonBeam.energy as BeamRadianceStandard looks it up */
                        /* (BeamRadiance.h:449-455)                              */
  float s_ext, e_ext, c_ext; /* PhotonSeg.s_ext/e_ext/c_ext (heterogeneous)      */
  int32_t seg_id;       /* PhotonSeg.seg_id                                      */
  int32_t id;           /* PhotonSeg.id (beam index)                             */
  uint32_t flags;       /* bit0 ada_width, bit1 scattered                        */
  uint32_t reserved[3]; /* [0]: 1 on segments pb_emit_beams / pb_recast_beams made from the
                           scattered list, else 0 (the renderer ignores it); [1..2] zero   */
} pb_segment;

/* One photon beam: PhotonBeam (CastPhotons.h:14-27) without the unused fields. */
typedef struct pb_beam {
  float start[4];
  float end[4];
  float omega_u[4];
  float radius;
  float energy[3];
  int32_t index_hit;
  uint32_t flags;       /* bit0 ada_width, bit1 scattered, bit2 absorbed,
                           bit3 pushed into beams_scattered (else beams_const)   */
  uint32_t photon;      /* emitting photon index (0-based)                       */
  uint32_t order;       /* push order inside that photon                         */
} pb_beam;

/* LBVH node as downloaded for checking: children >= 0 are internal nodes,
 * ~child (negative) is the leaf holding sorted position ~child. */
typedef struct pb_node {
  float lo[3];
  float hi[3];
  int32_t left, right;
} pb_node;

typedef struct pb_stats {
  uint64_t rays;             /* camera rays of the last pb_render call            */
  uint64_t rays_hit_scene;   /* rays for which ClosestIntersection found a hit   */
  uint64_t candidates;       /* (ray, segment) pairs the kernel evaluated         */
  uint64_t cyl_hits;         /* pairs with a non-negative discriminant            */
  uint64_t valid_hits;       /* pairs that contributed radiance                   */
  uint64_t nodes_visited;    /* LBVH nodes tested by the tile traversal           */
  uint64_t gate_passes;      /* (ray, segment) pairs whose root and gate boxes are
                                entered within the surface distance: the pairs on
                                which the reference calls HitCylinder
                                (BeamRadiance.h:428-439); counted by a full scan   */
  uint64_t shortcut_wrong;   /* pairs that the fast gate shortcut accepted and the exact
                                HitBoundingBox gate would have rejected (expected: 0)  */
  float ms_render;           /* device time of the last pb_render, or of the last pb_render_*_d with counters on
                                (HIP events; the device-pointer calls record none otherwise: two event packets cost
                                a small frame a tenth of its time)                */
  float ms_build;            /* device time of the last pb_build_lbvh             */
  float ms_prepare;          /* device time of the last segment upload/prepare    */
  float ms_emit;             /* device time of the last pb_emit_beams             */
} pb_stats;

typedef struct pb_ctx pb_ctx;

enum { PB_MODE_DEFAULT = 0,    /* what Control selects: GENERIC if adaptive, else 73HET / 73 / 721 by one_dimensional
                                  and heterogeneous (the dispatch of skeleton.cpp:225-262)                           */
       PB_MODE_721 = 721,      /* BeamRadianceStandard + Integral_721 (BeamRadiance.h:415-469, :86-110)             */
       PB_MODE_73HET = 73,     /* BVHBeamRadianceHet + Integral_73Het (:356-414, :211-288); needs pb_set_grid       */
       PB_MODE_73 = 730,       /* BeamRadiance with ONE_DIMENSIONAL in a homogeneous medium: Integral_73 (:510-524,
                                  :112-209 with HETEROGENEOUS false)                                                  */
       PB_MODE_GENERIC = 471 };/* BeamRadiance as written (:471-555): a segment with ada_width set is a cone (HitCone,
                                  Intersections.h:145-305, + Integral_721_ada, BeamRadiance.h:63-84), any other a
                                  cylinder with Integral_73 if control.one_dimensional, else Integral_721             */

/* Control.h:4-50 defaults. */
void pb_control_defaults(pb_control* c);

/* Human-readable text for the last failing call on this thread. */
const char* pb_last_error(void);

/* Number of usable GPUs; 0 if none (nothing here runs on the CPU). */
int pb_device_count(void);

/* Create/destroy a context on GPU `device`.  The Control values are copied. */
int pb_create(const pb_control* control, int device, pb_ctx** out);
void pb_destroy(pb_ctx* ctx);

/* The Cornell box of LoadTestModel (TestModelH.h:45-214) in world space:
 * writes 30 triangles as v0,v1,v2 xyz (9 floats each) and, if not NULL, their
 * normals (3 floats each).  Returns the triangle count (30), or PB_ERR_CAPACITY if capacity < 30. */
int pb_load_test_model(float* tri_xyz, float* normals_xyz, int capacity);

/* TransformationMatrix (skeleton.cpp:512-532): column-major 4x4. */
void pb_transformation_matrix(const float theta[3], const float camera[4], float matrix[16]);

/* The (matrix, triangles) pair that ClosestIntersection (Intersections.h:458-497)
 * receives on every call: triangles in world space, matrix as above. */
int pb_set_scene(pb_ctx* ctx, const float* tri_xyz, int n_tri, const float matrix[16]);

/* CreateSurface + UpdateHeight (Tessendorf.h:47-132): the (triangle_number+1)^2 points of
 * GRID.geometric_points as xyzw, z = |sum of the wave spectrum| + height.  Host arithmetic only;
 * returns the number of points, or a negative pb_status if capacity_points is too small. */
int pb_create_surface(int triangle_number, float height, double time, float* points_xyzw, int capacity_points);

/* matrix * point for n xyzw points in place, with glm's mat4 * vec4 operation order: the loop of
 * HeterogeneousRun (skeleton.cpp:119-125) that moves GRID into camera space. */
void pb_transform_points(const float matrix[16], float* points_xyzw, int n);

/* The extinction field that Extinction3D reads (Heterogeneous.h:152-163): GRID.geometric_points[i].z
 * after HeterogeneousRun's transform, side*side values, i = row * side + column, side = GRID.side_points.
 * 2 <= side <= 4096.  The first call with a given side in a process also checks on the host (about 50 ms)
 * that the kernels' two-multiply-add form of x / (2 / (side - 1)) rounds like the division for every x;
 * where it does not, they divide.  Results do not depend on the outcome. */
int pb_set_grid(pb_ctx* ctx, const float* sigma_t, int side);

/* Extinction3D (Heterogeneous.h:29-135) for n rays at once: start/end are xyz triples in camera space.
 * mean_out[i] is the function's return value (mean of the samples), count_out[i] (optional) the number
 * of (distance, extinction) samples, samples_out (optional) the first max_samples of them per ray as
 * float pairs.  Cell indices outside the grid are clamped (the reference reads out of bounds there). */
int pb_extinction3d(pb_ctx* ctx, const float* start_xyz, const float* end_xyz, int n, float* mean_out,
                    int* count_out, float* samples_out, int max_samples);

/* The segment set that BuildTree (CastPhotons.h:800-1126) would hold, and the
 * root box that CastPhotonBeams returns (CastPhotons.h:384-389, un-flipped).
 * Host and device-pointer forms.  Also derives the per-segment frame that
 * HitCylinder (Intersections.h:313-328) recomputes on every call. */
int pb_upload_segments(pb_ctx* ctx, const pb_segment* segs, int n,
                       const float root_lo[3], const float root_hi[3]);
int pb_upload_segments_d(pb_ctx* ctx, const pb_segment* d_segs, int n,
                         const float root_lo[3], const float root_hi[3], void* stream);
int pb_segment_count(pb_ctx* ctx, int* n_out);
int pb_download_segments(pb_ctx* ctx, pb_segment* out, int capacity, int* n_out);
int pb_get_root_box(pb_ctx* ctx, float root_lo[3], float root_hi[3]);

/* The segment set on disk: a 64-byte header ("PBSEGS\0\1", version 1, record size 112, count, root box) followed by
 * the pb_segment records (layout: raytracer_amd/csrc/pb_file.cpp).  It stands where the reference would hand
 * `segments` (CastPhotons.h:83) from one run to the next: a set dumped from one host - the reference's own tree
 * segments with their leaf boxes, say - renders on another without repeating the photon pass, and the same file is
 * the checkpoint of a stop-motion run.  The two file functions are host code and need no GPU; pb_save_segments
 * writes the context's current set and root box, pb_load_segments is pb_upload_segments from a file. */
int pb_write_segments_file(const char* path, const pb_segment* segs, int n, const float root_lo[3], const float root_hi[3]);
/* out may be NULL to ask for the count (and the root box) only. */
int pb_read_segments_file(const char* path, pb_segment* out, int capacity, int* n_out, float root_lo[3], float root_hi[3]);
int pb_save_segments(pb_ctx* ctx, const char* path);
int pb_load_segments(pb_ctx* ctx, const char* path);

/* CastPhotonBeams (CastPhotons.h:312-390) + BoundPhotonBeams (:144-288) on the
 * GPU: one photon per lane, counter-based random numbers keyed by
 * (seed, photon, draw), beams pushed in (photon, order) order.  The result
 * replaces the context's segment set (own-box cull rule) and root box. */
int pb_emit_beams(pb_ctx* ctx, int n_photons, uint64_t seed, int* n_beams_out, int* n_segments_out);
/* RecastPhotonBeams (CastPhotons.h:290-310) + BoundPhotonBeams as ProduceStopMotion uses them
 * (skeleton.cpp:312-324): every beam of the first list is cast again from its start along its direction
 * through the current medium (pb_set_grid), the scattered list is rebuilt from what that produces, the
 * first list is kept, and both are cut into segments again.  Random numbers: (seed, beam index, draw).
 * The reference's loop runs over the vector it appends to and does not terminate; exactly the beams
 * present on entry are re-cast here.  The root box is the bounds of the kept and the new beams. */
int pb_recast_beams(pb_ctx* ctx, uint64_t seed, int* n_scattered_out, int* n_segments_out);
int pb_download_beams(pb_ctx* ctx, pb_beam* out, int capacity, int* n_out);  /* const list, then scattered list */
int pb_beam_counts(pb_ctx* ctx, int* n_const, int* n_scattered);

/* BuildTree (CastPhotons.h:800-1126) replacement: Morton codes of the segment
 * midpoints, LDS radix sort, Karras hierarchy, node boxes from tabulated unions over each node's
 * range of the sorted order.  The kernels are queued on the context's own stream and not waited for; render calls
 * that traverse the tree on another stream wait for them there.  A build of a segment set that fits the buffers of an
 * earlier one allocates nothing. */
int pb_build_lbvh(pb_ctx* ctx);
/* keys[n], perm[n], nodes[n-1] (any may be NULL). */
int pb_download_lbvh(pb_ctx* ctx, uint64_t* keys, uint32_t* perm, pb_node* nodes);

/* DrawStandard / DrawHet (skeleton.cpp:330-365, :367-412) for the pixel
 * rectangle [x0,x0+w) x [y0,y0+h): per pixel ClosestIntersection, then the
 * beam radiance estimate over the segment set.  Output is current/SSAA per
 * pixel as RGB, row-major h*w*3 (float: the dvec3 -> vec3 conversion at
 * PutPixelSDL, SDLauxiliary.h:149).  use_lbvh selects how a tile finds its candidate segments:
 * 0 scans every segment, 1 traverses the LBVH of pb_build_lbvh, 2 bins the segments by their
 * projection (all camera rays share the origin) and sums in fixed point, order-independently. */
int pb_render(pb_ctx* ctx, int x0, int y0, int w, int h, int mode, int use_lbvh, float* rgb_out);
int pb_render_d(pb_ctx* ctx, int x0, int y0, int w, int h, int mode, int use_lbvh,
                float* d_rgb_out, void* stream);
/* Same for the 8-row strips first_strip + k*strip_stride (k = 0..n_strips-1) of
 * the full-width image, strip s covering rows 8s..8s+7: the interleaved row
 * tiles one GPU of several renders.  Output rows are packed: 8*n_strips rows. */
int pb_render_strips_d(pb_ctx* ctx, int first_strip, int strip_stride, int n_strips, int mode,
                       int use_lbvh, float* d_rgb_out, void* stream);

/* pb_render_d through the screen-space bins (use_lbvh = 2) for callers that cannot allocate: tight serving loops,
 * harnesses that forbid hipMalloc, code under HIP graph capture (there only the frames after the first of a given
 * segment set, rectangle and mode: the first one reads two counters back and waits for them on `stream`, which
 * capture does not allow - render it once before capturing).  It has no counterpart in the reference.  It calls no
 * hipMalloc, hipFree or other allocating routine: everything the frame needs besides its inputs (the context's
 * prepared segment set and scene) and its output lives in one file-scope device array of PB_NOALLOC_SCRATCH_BYTES
 * that every GPU the library runs on holds from the first pb_create on it until the process ends.  Frames beyond the
 * maxima below are refused with PB_ERR_CAPACITY and a message naming the limit; the result equals pb_render_d's bit
 * for bit (PB_MODE_73HET: pb_render_d's under pb_set_precise(ctx, 1) - the array has no room for the per-ray walk
 * tables of the relaxed-arithmetic estimator, so this entry point always runs the exact-order one; PB_MODE_73 runs the
 * relaxed estimator with its undecided pairs evaluated inside the items instead of in a launch of their own, which
 * gives pb_render_d's bits).  All contexts of a GPU share that array and it keeps the last call's contents (the call
 * clears what it needs zeroed), so calls that overlap in time on one GPU - other streams, other threads - race: issue them one
 * after the other.  The first frame of a given (segment set, rectangle, mode) waits once on `stream` to read two
 * counters back (how many list entries and evaluation items the frame has); later frames of it do not. */
#define PB_NOALLOC_MAX_GROUPS 4096          /* 8x8-pixel tiles x SSAA^2 of the rectangle: 512x512 at SSAA 1     */
#define PB_NOALLOC_MAX_LIST_ENTRIES 1048576 /* (64x64-pixel macro tile, segment) incidences of the frame        */
#define PB_NOALLOC_MAX_ITEMS 131072         /* evaluation items (tile x slice of its macro tile's list)         */
#define PB_NOALLOC_MAX_QUEUE_SLOTS 8192     /* 64 per 64x64-pixel macro tile the rectangle touches: 128 macro     *
                                             * tiles, which only a very wide and thin rectangle exceeds           */
#define PB_NOALLOC_SCRATCH_BYTES (28u << 20) /* the array: 28 MiB per GPU                                        */
int pb_render_noalloc_d(pb_ctx* ctx, int x0, int y0, int w, int h, int mode, float* d_rgb_out, void* stream);

/* Arithmetic of the per-pair estimate.  0 (default): fused multiply-adds, hardware reciprocal and
 * square root, entry/exit distances taken from the ray parameter; agrees with the reference to its own
 * float rounding noise (well inside max-abs 2e-3*max(1,|ref|), RMSE 1e-4).  1: every operation of
 * HitCylinder / Integral_721 in the reference's order with IEEE divide and sqrt (about 3x slower).
 * The gate (HitBoundingBox + max_distance) is exact in both.
 * PB_MODE_73 / PB_MODE_73HET through the screen-space bins (use_lbvh = 2): 0 takes the closest approach of beam and
 * view ray from the roots of the cylinder crossing in float and the view ray's mean extinction from a walk through
 * the grid done once per ray and frame; pairs whose bounds float arithmetic cannot decide safely (a beam seen
 * end-on, a closest approach at a cap, a grid corner crossed by two axes at once) are evaluated in the reference's
 * order in double, as every pair is under 1 (about 2x slower).  Same tolerance. */
int pb_set_precise(pb_ctx* ctx, int on);
/* The PB_TUNE_* environment switches (experiments and diagnostics, none needed in use) are read
 * when a context is created, never by a frame; a process that changes them while a context lives calls this to have
 * them read again.  No counterpart in the reference (its switches are #defines, Control.h:4-31). */
void pb_reload_tuning(void);

/* PutPixelSDL (SDLauxiliary.h:149-161): RGB float -> 0x80RRGGBB, n pixels. */
int pb_quantise_argb(const float* rgb, int n_pixels, uint32_t* argb_out);

/* Counters and device times of the last calls. Counting slows pb_render, so it
 * is off unless enabled. */
int pb_enable_counters(pb_ctx* ctx, int on);
int pb_counters(pb_ctx* ctx, pb_stats* out);

/* ---- known-answer probes --------------------------------------------------------------------------
 * One GPU lane per query through the very device functions the frame and photon kernels inline; host
 * arrays in and out.  For tests against vectors made with the reference's own functions; not used by the
 * render path.
 *
 * ClosestIntersection (Intersections.h:458-497) for n rays (start, dir as xyzw) against the scene of
 * pb_set_scene: found flag, Intersection.position (xyzw), .distance, .index (zeros / -1 where not found). */
int pb_probe_closest_intersection(pb_ctx* ctx, const float* start_xyzw, const float* dir_xyzw, int n,
                                  uint8_t* found, float* pos_xyzw, float* dist, int32_t* index);
/* HitBoundingBox (Intersections.h:81-143) for rays from (0,0,0) (every camera ray) against boxes given
 * un-flipped (lo = min_b, hi = max_b of :85-86): hit flag and |hit - start| as BeamRadiance.h:429 forms it. */
int pb_probe_hit_bounding_box(pb_ctx* ctx, const float* lo_xyz, const float* hi_xyz, const float* dir_xyzw, int n,
                              uint8_t* hit, float* entry_distance);
/* Pair i = (ray from (0,0,0) along dir[i], segs[i]) with its own gate box and no surface in the way.
 * cyl_out12[i] = HitCylinder's (Intersections.h:307-456) return value, valid, tb_minus, tb_plus, tc_minus,
 * tc_plus, entry_point, exit_point (zeros where the reference leaves them unset) in the exact operation
 * order; integral[i], status[i] = value of Integral_721 (BeamRadiance.h:86-110; fast != 0: the default
 * fast form) or Integral_73Het (:211-288) and 0 no hit / 1 hit without contribution / 2 contributed. */
int pb_probe_pairs(pb_ctx* ctx, const pb_segment* segs, const float* dir_xyzw, int n, int mode, int fast,
                   float* cyl_out12, double* integral, int32_t* status);

/* RunAnalysis (Analysis.h:81-122): the estimator variance study for the reference's fixed geometry
 * (SetUpAnalysis, :455-475), one GPU lane per kernel width.  table 0 = NSD.csv: normalised standard
 * deviation of BsBs2D1, BsBs1D, BsBl2D2, PBl3D, BsBl1D, BlBs1D, BlP3D, PBs2D (:94-101); table 1 =
 * rRMSE.csv: rBsBs2D1, rBsBl1D, rBlP3D (:111-117).  extinction = analysis_extinction (:15; the
 * reference's is EXT_ANALYSIS = 0.04f widened to double).  Row r: first_col[r] = kernel_width / mfp,
 * values[r*cols + k], present[r*cols + k] = 0 where the reference writes an empty cell.  *n_rows is set
 * even when capacity_rows is too small (PB_ERR_CAPACITY). */
int pb_run_analysis(pb_ctx* ctx, double extinction, int table, double* first_col, double* values,
                    uint8_t* present, int capacity_rows, int* n_rows);

#ifdef __cplusplus
}
#endif
#endif /* PHOTON_BEAMS_H */
