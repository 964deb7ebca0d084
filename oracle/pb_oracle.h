/* TEST INFRASTRUCTURE — CPU restatement of the reference's photon-beam path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (libphoton_beams.so) never does.
 *
 * Parity pin: the full configuration-1 pipeline of this restatement
 * (CastPhotonBeams -> BoundPhotonBeams -> BuildTree -> DrawStandard ->
 * PutPixelSDL) is checked against the reference's committed render
 * (screenshot.png/.bmp) in tests/test_oracle_golden.py, and ClosestIntersection,
 * HitBoundingBox, HitCylinder, HitCone, findRotationMatrix, LoadTestModel and
 * CreateSurface are checked bit-for-bit against the reference's own functions
 * compiled from Source/Intersections.h, TestModelH.h and Tessendorf.h
 * (oracle/_ref).  RunAnalysis (analysis_oracle.cpp) reproduces the reference's
 * committed NSD.csv and the rRMSE.csv its own function writes, text for text.
 * The heterogeneous branch (Extinction3D, the het paths of CastBeam,
 * BoundPhotonBeams and BuildTree, BVHBeamRadianceHet, Integral_73Het) is pinned
 * by the numbers the unmodified reference produced for HETEROGENEOUS +
 * ONE_DIMENSIONAL, 10 000 photons, BOUNCES 0, SEED 42: 7 370 + 341 beams,
 * 203 559 segments, 149 629 in the tree, 42 014 lost - all reproduced exactly -
 * and its image: mean 94.8/255 and 69 247 saturated pixels reproduced exactly,
 * 236 536 non-zero pixels against 236 506 (the reference reads the grid out of
 * bounds for a few beams, the restatement clamps); tests/
 * test_oracle_het_reference_numbers.py.  That is a pin through counts and image
 * statistics, not value by value: no per-call output of these functions exists.
 * The functions whose headers need SDL and that no reference run covers -
 * Integral_73 (homogeneous), Integral_721_ada, the SHORT_VIEW draw order and
 * RecastPhotonBeams (which does not terminate as written; see there) - are
 * restated only: parity UNPINNED.
 */
#ifndef PB_ORACLE_H
#define PB_ORACLE_H
#include <stdint.h>
#include "../include/photon_beams.h" /* data formats only: pb_control, pb_segment, pb_beam */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct orc_counters {
  uint64_t rays, rays_hit_scene, node_visits, cyl_calls, cyl_true, valid_hits;
} orc_counters;

/* ---- primitives (known-answer tests against oracle/_ref) ---- */
int  orc_load_test_model(float* tri_v0v1v2_xyzw /*n*12*/, float* normals_xyzw /*n*4*/, int capacity);
void orc_transformation_matrix(const float theta[3], const float camera[4], float m16[16]);
int  orc_closest_intersection(const float start[4], const float dir[4], const float m16[16],
                              const float* tri_xyzw, int n_tri, float position[4], float* distance, int* index);
int  orc_hit_bounding_box(const float box_min[4], const float box_max[4], const float start[4],
                          const float dir[4], float hit[4]);
/* out[0]=returned bool, out[1]=valid, then tb-,tb+,tc-,tc+, entry xyz, exit xyz (12 floats) */
void orc_hit_cylinder(const float start[4], const float dir[4], const float seg_start[4],
                      const float seg_end[4], const float seg_orig_start[4], float radius, float out[12]);
/* HitCone (Intersections.h:145-305), same output layout */
void orc_hit_cone(const float start[4], const float dir[4], const float seg_start[4],
                  const float seg_end[4], const float seg_orig_start[4], float radius, float out[12]);
/* Integral_721_ada (BeamRadiance.h:63-84) */
double orc_integral_721_ada(const float seg_start[4], const float seg_end[4], const float seg_orig_start[4],
                            float radius, float tb_minus, float tc_minus, float tc_plus, float extinction,
                            float scattering, const float dir[4]);
/* RunAnalysis (Analysis.h:81-122): table 0 = NSD.csv (8 columns), 1 = rRMSE.csv (3 columns); returns the row count */
int orc_run_analysis(double extinction, int table, double* first_col, double* values, uint8_t* present, int capacity);
void orc_find_rotation_matrix(const float c_dir[3], const float w_dir[3], float m9[9]);
double orc_integral_721(const float seg_start[4], const float seg_end[4], float tb_minus, float tc_minus,
                        float tc_plus, float extinction, const float dir[4]);
int  orc_create_surface(int triangle_number, float height, double time, float* points_xyzw /*(n+1)^2*4*/);
/* returns number of samples written (<= capacity); mean returned through *mean */
int  orc_extinction3d(const float* grid_sigma /*side*side*/, int side, const float start[4], const float end[4],
                      float* dist_ext /*capacity*2*/, int capacity, float* mean);
double orc_integral_73het(const float* grid_sigma, int side, const pb_segment* seg, const float entry[3],
                          const float exit_[3], float extinction_c, float scattering_c, const float dir[4]);
void orc_put_pixel(const double rgb[3], uint32_t* argb);

/* ---- pipeline ---- */
orc_ctx* orc_create(const pb_control* control);
void orc_destroy(orc_ctx*);
/* CastPhotonBeams with std::mt19937(seed) (rng_kind 0) or the counter-based
 * generator the GPU emission uses (rng_kind 1). Returns beams_const.size()+beams_scatt.size(). */
int  orc_cast_photon_beams(orc_ctx*, int number, int rng_kind);
int  orc_recast_photon_beams(orc_ctx*, int rng_kind, uint64_t seed);   /* rebuilds the scattered list; returns its size */
int  orc_beam_counts(orc_ctx*, int* n_const, int* n_scatt);
int  orc_get_beams(orc_ctx*, pb_beam* out, int capacity);   /* const list then scattered list */
int  orc_bound_photon_beams(orc_ctx*);                       /* both lists; returns items.size() */
int  orc_build_tree(orc_ctx*, int max_depth);                /* returns tree segment count, <0 on runaway */
int  orc_tree_stats(orc_ctx*, int* n_nodes, int* n_leaves, int* max_depth);
int  orc_lost_segments(orc_ctx*);                            /* size of lost_segments after BuildTree (het branch) */
void orc_root_box(orc_ctx*, float lo[3], float hi[3]);       /* un-flipped */
/* which: 0 = items with their own box, 1 = tree segments with their leaf-node box */
int  orc_export_segments(orc_ctx*, int which, pb_segment* out, int capacity);
int  orc_set_grid(orc_ctx*, const float* sigma, int side);
int  orc_make_grid(orc_ctx*, int triangle_number, float height, double time, float* sigma_out /*side*side or NULL*/);
/* DrawStandard/DrawHet by recursive traversal of the restated tree (reference semantics). */
int  orc_render_tree(orc_ctx*, int x0, int y0, int w, int h, int mode, double* rgb /*h*w*3*/,
                     orc_counters* counters, int threads);

/* The flat form of the estimate that the GPU implements: a segment contributes iff
 * the root box and its own cull box are entered within the surface distance. */
int  orc_render_segments(const pb_control* control, const float* tri_xyz /*n*9 world*/, int n_tri,
                         const float m16[16], const float* grid_sigma, int side,
                         const pb_segment* segs, int n, const float root_lo[3], const float root_hi[3],
                         int x0, int y0, int w, int h, int mode, double* rgb, orc_counters* counters,
                         int threads);
int  orc_render_segments_strided(const pb_control* control, const float* tri_xyz, int n_tri, const float m16[16],
                         const float* grid_sigma, int side, const pb_segment* segs, int n,
                         const float root_lo[3], const float root_hi[3], int x0, int y0, int w, int h,
                         int step_x, int step_y, int mode, double* rgb, orc_counters* counters, int threads);
/* CPU twin of the GPU's linear BVH: Morton keys (sorted), permutation, inner nodes. */
int  orc_lbvh(const pb_segment* segs, int n, uint64_t* keys_out, uint32_t* perm_out, pb_node* nodes_out);
int  orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
