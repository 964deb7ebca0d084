// Internal device/host structures of libphoton_beams.so.
#ifndef PB_INTERNAL_H
#define PB_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/photon_beams.h"
#include "pb_device_math.h"

// Tessendorf.h:4 — the reference's PI.
#define PB_PI_REF 3.14159265
// Tessendorf.h:23
#define PB_ACTUAL_WIDTH 2.0f

// One Cornell-box triangle in camera space, with the ray-independent cofactors
// of ClosestIntersection's inverse(mat3(-dir, e1, e2)) (Intersections.h:468-485).
// 16 floats = 64 B, read through scalar loads (the index is wave-uniform).
struct TriPrep {
  float v0[3];
  float e1[3];
  float e2[3];
  float c0, c1, c2;   // e1y*e2z-e2y*e1z, e1x*e2z-e2x*e1z, e1x*e2y-e2x*e1y
  float n[3];         // Triangle::normal.xyz (TestModelH.h:29-38), world == camera for theta = 0
  float pad;
};
static_assert(sizeof(TriPrep) == 64, "TriPrep layout");

// Everything HitCylinder (Intersections.h:313-346) derives from a segment alone,
// computed once per segment with the reference's operation order, plus the
// gate box and the radiance weight.  48 dwords = 192 B, 64-byte aligned so a
// wave-uniform record is fetched with s_load_dwordx16.  The first 32 dwords are
// all the default (fast) pair evaluation reads: two loads, one round trip.
struct __attribute__((aligned(256))) SegFrame {
  float Rinv[9];   //  0.. 8  R_inv, m[col][row] at col*3+row
  float ox2, oz2;  //  9,10   2*T_inv.x, 2*T_inv.z
  float c4;        // 11      4*c, c of Intersections.h:344-346
  float oy;        // 12      T_inv.y (T_inv == origin_prime for a camera ray from (0,0,0))
  float len;       // 13      |end-start|
  float tb0;       // 14      |start-orig_start|
  uint32_t qflags; // 15      bit i: origin inside slab i (MIDDLE); bit 3: origin inside the box;
                   //         bit 4: the gate box contains the segment's cylinder; bit 5: the record is a cone's
                   //         (HitCone, Intersections.h:145-305): frame about orig_start, c4 = 4 c of :196-200;
                   //         bit 6: the three weights W are equal
  double W[3];     // 16..21  energy*(1/(4 PI))*(sigma_s/(PI r^2)) (721) or energy*(1/(4 PI)) (73het)
  float lo[3];     // 22..24  gate box, min_b of HitBoundingBox
  float hi[3];     // 25..27  max_b
  float cand[3];   // 28..30  candidate planes of HitBoundingBox for the origin (0,0,0)
  float vcap10;    // 31      most a crossing's integral may be, times 2^-10: PB_FIX_LIMIT / max(W) scaled likewise
  float R[9];      // 32..40  inverse(R_inv) (Intersections.h:443-444)
  float ox, oz;    // 41,42   T_inv.x, T_inv.z
  float bdir[3];   // 43..45  normalize(end-start)
  float radius;    // 46
  float ymin;      // 47      cone records: y of seg.start in the cone's frame (Intersections.h:175); else 0
  float pad[16];   // 48..63  a record is two 128-byte L2 lines of its own: the halves the pair loop fetches (dwords 0..15 and
                   //         16..31, two scalar loads) always share the first one - at 192 bytes every other record straddled two
};
static_assert(sizeof(SegFrame) == 256, "SegFrame layout");
constexpr int PB_FRAME_USED_BYTES = 192;

// What Integral_73Het (BeamRadiance.h:211-288) reads of a segment beyond its end points: 8 dwords.
struct __attribute__((aligned(32))) HetFrame {
  float os[3];     // PhotonSeg.orig_start
  float s_ext;     // extinction at the segment's start / end sample (CastPhotons.h:271-272)
  float e_ext;
  float c_ext;     // running mean up to the segment (CastPhotons.h:190-194)
  int seg_id;      // 1-based index of the segment in its beam (CastPhotons.h:195)
  float radius;
};
static_assert(sizeof(HetFrame) == 32, "HetFrame layout");

// What the relaxed-arithmetic Integral_73 / Integral_73Het reads of a segment beyond its cylinder frame, in that
// frame's units (lengths divided by |end - start|; start is its origin, the axis its y): 8 dwords, one scalar load.
//   orig_start sits at (x, osy, z) with x^2 + z^2 = r2; aos / bos are its distances from start and end;
//   the beam's extinction at distance beta from orig_start is A + B (beta - aos) (BeamRadiance.h:262-266 with
//   proportion = (beta - aos) / (bos - aos)); krad = (sigma_s / sigma_t) / radius (:283-285).
struct __attribute__((aligned(32))) HetFast {
  float r2, osy, aos, bos, A, B, krad, pad;
};
static_assert(sizeof(HetFast) == 32, "HetFast layout");

// What Integral_73Het computes from the segment alone, in double: the unit axis, its dot product with itself, and
// the distances of the two end points from orig_start.  Filled once per prepared segment set (beam_axis in
// pb_pair.h) so that the per-hit code does not repeat three square roots and three divisions.
struct BeamAxis {
  double ab[3];
  double u_dot_u, a_os, b_os;
};
static_assert(sizeof(BeamAxis) == 48, "BeamAxis layout");

// A cone record's slot in the same array: cos^2 and sin^2 of HitCone's tan_theta (Intersections.h:184-200),
// each the square, in double, of the float cosine / sine.
struct __attribute__((aligned(32))) ConeFrame {
  double c2, s2;
  float pad[4];
};
static_assert(sizeof(ConeFrame) == 32, "ConeFrame layout");

// LBVH internal node as the tile traversal reads it: both child boxes inline, 64 B.
// child >= 0: internal node index; child < 0: leaf, ~child = position in Morton order.
struct __attribute__((aligned(64))) LbvhNode {
  float lo0[3], hi0[3], lo1[3], hi1[3];
  int child0, child1;
  int pad[2];
};
static_assert(sizeof(LbvhNode) == 64, "LbvhNode layout");

// Gate box of the tree root (BeamRadiance.h:426-430 applied to `root`).
struct GateBox {
  float lo[3], hi[3], cand[3];
  uint32_t qflags;
};

struct RenderParams {
  const TriPrep* tris;
  const float4* tri_box;   // per triangle: bounds of x/z, y/z seen from the camera (lo, hi, lo, hi)
  const float* tri_edge;   // per triangle 12 floats: its edges as lines a u + b v + c in the (x/z, y/z) plane, [9] = usable
  int n_tri;
  const SegFrame* frames;
  const float4* capA;      // (start.xyz, radius)
  const float4* capB;      // (end.xyz, 0)
  int n_seg;
  GateBox root;
  int width, height, ssaa; // SCREEN_WIDTH, SCREEN_HEIGHT, SSAA
  float focal;             // rasteriser.h:17
  float extinction_c;      // Control.h:48
  float scattering_c;      // Control.h:47
  const HetFrame* het;     // per segment: Integral_73/73Het inputs, or a ConeFrame where qflags bit 5 is set
  const BeamAxis* het_axis; // per segment: beam_axis() of its end points (unused values for a cone record)
  const HetFast* het_fast;  // per segment, PB_MODE_73 / PB_MODE_73HET: what the relaxed-arithmetic estimator reads
  float* walk;              // [groups][VIEW_WALK_STRIDE][64] the view rays' tabulated walks through the grid, or null
  uint2* residue;           // pairs the relaxed heterogeneous estimator leaves to the exact-order code and an item did not fill a
                            // wave with: (segment, ray group * 64 + ray) in PB_RESIDUE_QUEUES queues; null: every item evaluates its own
  int residue_capacity;
  int debug;                // PB_DEBUG_FLAGS: timing experiments that leave work out (wrong frames); 0 in normal use
  int one_dimensional;     // ONE_DIMENSIONAL, read by PB_MODE_GENERIC
  int short_view;          // SHORT_VIEW
  uint32_t view_seed;      // Control seed: key of the camera rays' draws
  const float* grid_z;     // sigma_t field (side*side) or nullptr
  int grid_side;
  int grid_fast_div;       // GridView::fast_div
  // geometry of this launch: tile (tx, k) covers pixels x = x0 + 8*tx + lx,
  // y = y0 + 8*k*strip_stride + ly; output row = 8*k + ly, row length out_w
  int x0, y0, out_w, out_h, strip_stride, tiles_x, tiles_y;
  float* out;              // out_h * out_w * 3
  unsigned long long* counters;  // 8 words or nullptr
  // workspace of one frame (see pb_render.hip)
  float* rays;              // [groups][PB_RAY_WORDS][64]: l.xyz, dn.xyz, maxd, 0, d1.xyz, 0
  int* group_last;          // [groups]: last chunk of the group's list, -1 if none
  uint32_t* chunks;         // [chunk_capacity][PB_CHUNK_WORDS]: group, count, prev, 0, indices
  double* partials;         // [chunk_capacity][3][64]
  uint32_t* ctrl;           // [0] chunks produced, [1] chunks taken by eval waves
  int chunk_capacity;
  int precise;              // 1: evaluate pairs in the reference's exact operation order
  // screen-space bins (pb_bins.hip)
  uint32_t* tile_count;     // [tiles] length of the tile's macro list
  uint32_t* sub_count;      // [macro tiles][PB_BIN_SUBS] candidates per macro tile and sub-counter
  uint32_t* sub_fill;       // [macro tiles][PB_BIN_SUBS] fill cursors of the second binning pass
  uint32_t* sub_off;        // [macro tiles][PB_BIN_SUBS] start of each sub-list in bin_list (a macro tile's are adjacent)
  uint32_t* bin_list;       // [bin_capacity] segment indices, grouped by tile
  uint32_t* blk;            // [binning blocks][macro tiles] lane-per-macro-tile binning: a block's count per macro tile, turned
                            // into the start of its range inside that macro tile's list by queue_build_kernel
  unsigned long long* bin_mask;   // [n_seg][PB_BIN_FAN] what the counting pass of the rectangle walk found (bit 63: valid), or null
  uint32_t* items;          // [items][4]: group, start, n, refined: 0 = n candidates at bin_list + start, still to be tested against
                            // the tile's pyramid; 1 = n survivors of that test at surv_list + start
  uint32_t* surv_list;      // the items' survivors, item after item in queue order (kept with the item queue)
  uint32_t* surv_cnt;       // [items] survivors per item, then [items] their offsets (while the lists are made)
  uint32_t* item_off;       // [queue slots] first item of the tile in that slot (macro tiles in rank order)
  uint32_t* macro_cnt;      // [macro tiles] list length
  uint32_t* macro_rank;     // [macro tiles] position in the evaluation queue, longest list first
  uint32_t* tile_items;     // [tiles] evaluation items of the tile (written with the items, kept with them)
  uint32_t* tile_done;      // [tiles] items of the tile that have added their sums this frame (finalize_in_eval)
  int xcd_run;              // evaluation items per run that one XCD takes (0: items are dealt singly round the XCDs)
  int store_d1;             // ray set-up also stores the once-normalised direction (estimators other than Integral_721)
  int finalize_in_eval;     // 1: the last item of a tile writes its pixels; tiles without items are written by ray set-up
  unsigned long long* accum;// [tiles][PB_ACC_PLANES][64] fixed-point radiance sums; a pixel's channel = its plane + the common plane
  int bin_capacity, item_capacity;
  int macro_w;              // tile columns per macro tile
  int macro_h;              // tile rows per macro tile (fewer when the rows are interleaved strips)
  int fused_queue;          // 1: ray set-up clears the accumulators and one block builds the queue (small frames)
  int n_items_host;         // number of items of this frame as measured when the workspace was sized
  int inline_item_blocks;   // > 0: the frame is one launch whose items set up their own rays (host: a request; kernel: the number
                            // of blocks that evaluate items, the blocks behind them blacken the tiles without items)
  int kept_queue;           // items is the kept queue of survivors (n_items_host of them; ctrl[0] still counts the queue as built)
  int slice;                // candidates of a macro list per evaluation item (<= PB_SLICE)
  int slice_whole;          // a list of at most this many candidates is one item whatever `slice` (<= PB_SLICE; 0: none)
  int eval_waves;           // size of the persistent evaluation grid
  // LBVH (lbvh_n < 2: scan all segments)
  const LbvhNode* lbvh_nodes;
  const float4* lbvh_sA;    // capsules in Morton order
  const float4* lbvh_sB;
  const uint32_t* lbvh_perm;
  int lbvh_n;
};

constexpr int PB_RAY_WORDS = 12;       // floats stored per camera ray
constexpr int PB_ACC_PLANES = 4;       // pixel sums per tile: red, green, blue, and what is common to all three
constexpr int PB_CHUNK = 124;          // candidates per chunk
constexpr int PB_CHUNK_WORDS = 128;    // 4 header words + PB_CHUNK indices = 512 B
constexpr int PB_EVAL_WAVES = 256 * 32;
constexpr int PB_MACRO = 8;             // most tiles a macro tile is wide or tall (RenderParams::macro_w x macro_h)
constexpr int PB_SLICE = 512;           // candidates of a macro list handed to one evaluation item
constexpr int PB_BIN_FAN = 8;            // lanes that share the macro rows of one segment in the binning passes
constexpr int PB_BIN_SUBS = 1;          // sub-counters per tile in the binning passes (power of two)
constexpr int PB_STACK_CAP = 2048;
// Fixed-point radiance accumulation of the binned path: a contribution c >= 0 is scaled by 2^-10 and added to
// PB_FIX_MAGIC = 4.0 in double, which rounds it to a multiple of ulp(4.0) = 2^-50 (2^-40 of c, 9e-13), and the
// bit patterns are summed as 64-bit integers, so the sum does not depend on the order in which candidates
// are met.  Valid for c < 4096.  (4.0 is an inline constant of the FP64 instructions; any other anchor costs
// two register moves per contribution.)
#define PB_FIX_MAGIC 4.0
#define PB_FIX_PRESCALE 0.0009765625          /* 2^-10 */
#define PB_FIX_SCALE 9.094947017729282e-13   /* 2^-40: one unit of the sums, in radiance */
#define PB_FIX_LIMIT 2047.0                   /* larger contributions saturate here (display range is [0,1]) */

enum { PB_CNT_RAYS = 0, PB_CNT_RAYS_HIT, PB_CNT_CAND, PB_CNT_CYL, PB_CNT_VALID, PB_CNT_NODES, PB_CNT_GATED, PB_CNT_SHORTCUT_WRONG,
       // evaluation-queue timing of a counted frame, 100 MHz ticks: ~first wave start, ~first wave that found the
       // queue empty (both stored as ~t so that atomicMax yields the minimum), last wave end; items processed
       PB_CNT_T_START_INV, PB_CNT_T_DRAIN_INV, PB_CNT_T_END, PB_CNT_ITEMS, PB_CNT_N };

// Device memory one frame works in besides its inputs and its output (see pb_render.hip).  A context owns one
// that grows on demand (hipMalloc); pb_render_noalloc_d uses one of fixed capacities laid out in a file-scope
// device array, shared by every context on that GPU.
struct FrameWorkspace {
  bool fixed = false;                  // capacities cannot grow: a need beyond them is PB_ERR_CAPACITY
  float* d_rays = nullptr;
  int* d_group_last = nullptr;
  uint32_t* d_chunks = nullptr;
  double* d_partials = nullptr;
  uint32_t* d_ctrl = nullptr;          // 4 control words (items / chunks produced, list entries)
  int ws_groups = 0, ws_chunks = 0;
  uint32_t* d_tile_u32 = nullptr;      // tile_count | item_off | macro_cnt | macro_rank (ws_tiles each) | sub_count | sub_fill | sub_off
  uint32_t* d_bin_list = nullptr;
  uint32_t* d_blk = nullptr;           // per (binning block, macro tile) counts / range starts
  size_t ws_blk_cap = 0;
  uint32_t* d_items = nullptr;
  unsigned long long* d_accum = nullptr;
  uint32_t* d_scan_tmp = nullptr;
  float* d_walk = nullptr;             // per ray group: the view rays' tabulated grid walks (heterogeneous medium only)
  int ws_walk_groups = 0;
  uint32_t* d_surv_list = nullptr;     // survivors of every item's candidates (RenderParams::surv_list), kept with the queue
  size_t ws_surv_cap = 0;
  uint32_t* d_surv_cnt = nullptr;      // per item: survivor count, offset; per queue slot: new items, their offset; scan scratch
  size_t ws_surv_words = 0;            // words d_surv_cnt holds
  uint32_t* d_items_kept = nullptr;    // the kept queue: items of survivors (RenderParams::items of the kept frames)
  int ws_items_kept_cap = 0, ws_items_kept = 0;
  bool surv_ready = false;             // d_items_kept / d_surv_list / tile_items describe the frame of ws_key
  bool sums_left_clear = false;        // the last frame ended with every multi-item tile's sums and arrival counter at zero
  uint2* d_residue = nullptr;          // device-wide queue of pairs for the exact-order code (relaxed heterogeneous estimator)
  int ws_residue_cap = 0;
  int ws_tiles = 0, ws_bin_cap = 0, ws_item_cap = 0, ws_items_needed = 0;
  size_t ws_accum_px = 0;
  unsigned long long ws_key[6] = {0, 0, 0, 0, 0, 0};  // segment set + frame geometry the capacities were verified for
  unsigned long long owner = 0;        // id of the context whose frame ws_key describes (fixed workspaces are shared)
  int ws_slice = 0;                    // item length chosen for the frame geometry of ws_key (0: not chosen yet)
  bool queue_kept = false;             // list offsets and item queue of ws_key's frame are in place (fused queue only)
};

void pb_set_error(const std::string& msg);

// The PB_TUNE_* / PB_DEBUG_* switches (diagnostics and experiments; none is needed in use), read from the environment
// when the first context is created and again by every pb_create and pb_reload_tuning - never by a frame.
struct PbTune {
  int surv = -1, fused_max = -1, macrow = -1, macroh = -1, xcdrun = -1, whole = -1, slice = -1, waves = -1, raywaves = -1, debug = 0;
  double inline_ratio = -1.0;
  bool no_prefine = false, no_keep = false, separate_finalize = false, refill = false, debug_times = false, emit_twice = false,
       het1 = false, het2 = false, old_eval = false, no_residue = false, wide = false, het_lds_list = false, het1_lds_tab = false;
};
const PbTune& pb_tune();
void pb_tune_reload();

// One GPU's state (opaque to callers of the C ABI).
struct pb_ctx {
  pb_control ctl;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};   // pairs: 0,1 render; 2,3 prepare; 4,5 build; 6,7 emit
  bool ev_valid[4] = {false, false, false, false};
  // Ordering between calls that are queued on different streams and not waited for.  Frames read the segment
  // records, the scene, the grid and the LBVH on the stream the caller names; the calls that rewrite those buffers run
  // on a stream of their own choosing (upload) or on the context's (photon pass, LBVH build, grid, scene).  A frame
  // costs nothing for this: it only notes its stream here.  A writer records ev_readers on every noted stream when it
  // starts - which covers every frame queued there before - and makes its own stream wait for it.
  hipStream_t reader_streams[8] = {};
  int n_reader_streams = 0;
  bool readers_overflow = false;        // more than eight distinct streams: writers wait for the whole device
  hipEvent_t ev_readers = nullptr;      // hipEventDisableTiming
  hipStream_t prepare_stream = nullptr; // stream of the last pb_prepare_frames (ev[3] marks its end)
  // scene
  float* d_tri_xyz = nullptr;
  float* d_matrix = nullptr;
  TriPrep* d_tris = nullptr;            // followed by n_tri float4 screen boxes and n_tri x 12 floats of screen edges
  int n_tri = 0;
  float matrix[16];
  // beams of the last pb_emit_beams: beams_const list, then beams_scattered list
  pb_beam* d_beams = nullptr;
  size_t cap_beams = 0;                    // records d_beams can hold
  pb_beam* d_beams_spare = nullptr;        // the other of the two beam buffers: a new list never aliases the old one
  size_t cap_beams_spare = 0;
  uint32_t* d_emit_tmp[3] = {nullptr, nullptr, nullptr};   // photon-pass scratch (counts, offsets, scan space; beam staging), kept between calls
  size_t emit_tmp_words[3] = {0, 0, 0};
  int n_beams_const = 0, n_beams_scatt = 0;
  // segments
  pb_segment* d_segs = nullptr;
  SegFrame* d_frames = nullptr;
  float4* d_capA = nullptr;
  float4* d_capB = nullptr;
  HetFrame* d_het = nullptr;
  BeamAxis* d_het_axis = nullptr;
  HetFast* d_het_fast = nullptr;
  // extinction field of the heterogeneous medium
  float* d_grid = nullptr;
  size_t grid_cap_floats = 0;      // floats d_grid holds: a new field of the same side (stop motion) reuses the buffer
  int grid_side = 0;
  int grid_fast_div = 0;   // the cell size passed pb_set_grid's division check (GridView::fast_div)
  int n_seg = 0, cap_seg = 0, frame_mode = 0;
  bool have_segments = false;
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
  // output staging for the host-pointer render
  float* d_out = nullptr;
  size_t out_floats = 0;
  // LBVH of the segment set (pb_lbvh.hip)
  uint64_t* lb_keys = nullptr;
  uint32_t* lb_perm = nullptr;
  LbvhNode* lb_nodes = nullptr;
  float4* lb_sA = nullptr;
  float4* lb_sB = nullptr;
  int* lb_left = nullptr;
  int* lb_right = nullptr;
  float* lb_node_box = nullptr;
  float* lb_leaf_box = nullptr;
  void* lb_pool = nullptr;                 // the one allocation all of the above and the build's scratch live in
  size_t lb_cap = 0;                       // segments it is laid out for
  uint64_t* lb_keys2 = nullptr; uint32_t* lb_vals2 = nullptr; uint32_t* lb_hist = nullptr; uint32_t* lb_scan_tmp = nullptr;
  int* lb_range_lo = nullptr; int* lb_range_hi = nullptr; float* lb_unions = nullptr; uint32_t* lb_bounds = nullptr;
  int lb_n = -1;
  unsigned long long lb_version = ~0ull;
  // frame workspace: grown on demand by this context's render calls
  FrameWorkspace ws;
  uint32_t* h_ctrl = nullptr;          // pinned copy of the workspace's control words
  unsigned long long id = 0;           // unique per created context (a freed context's address may be handed out again)
  unsigned long long segs_version = 0; // bumped whenever the segment set or scene changes
  // counters
  unsigned long long* d_counters = nullptr;
  bool count_on = false;
  bool time_render = false;               // record the frame's HIP events although counters are off (host-pointer pb_render)
  bool precise = false;
  int tune_slice = 0, tune_waves = 0;     // 0: defaults chosen per frame   // PB_TUNE_SLICE / PB_TUNE_WAVES environment overrides
  pb_stats stats;
};

int pb_ensure_segment_capacity(pb_ctx* ctx, int n);
// A call about to overwrite what frames read, on stream `writer`: `writer` waits for every frame queued so far.
// host_waits: the call ends with a host wait on `writer`, after which no noted frame is still running.
int pb_wait_for_readers(pb_ctx* ctx, hipStream_t writer, bool host_waits);
void pb_note_reader(pb_ctx* ctx, hipStream_t s);
int pb_prepare_frames(pb_ctx* ctx, int mode, hipStream_t s);
int pb_default_mode(const pb_ctx* ctx);
int pb_emit_run(pb_ctx* ctx, int n_photons, uint64_t seed, bool recast);
int pb_lbvh_build(pb_ctx* ctx);
int pb_lbvh_download(pb_ctx* ctx, uint64_t* keys, uint32_t* perm, pb_node* nodes);

#define PB_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      pb_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                       \
      return PB_ERR_DEVICE;                                                                  \
    }                                                                                        \
  } while (0)

// launchers implemented in the kernel translation units
int pb_launch_prepare_scene(const float* d_tri_xyz, int n_tri, const float* matrix16, TriPrep* d_out,
                            float4* d_screen_box, float* d_screen_edge, hipStream_t s);
int pb_launch_prepare_segments(const pb_segment* d_segs, int n, int mode, int one_dimensional, float scattering_c, float extinction_c,
                               SegFrame* d_frames, float4* d_capA, float4* d_capB, HetFrame* d_het, HetFast* d_het_fast, hipStream_t s);
int pb_launch_beam_axis(const float4* d_capA, const float4* d_capB, const HetFrame* d_het, int n, BeamAxis* d_out, hipStream_t s);
int pb_launch_cull(const RenderParams& p, int mode, bool count, hipStream_t s);
int pb_launch_eval_reduce(const RenderParams& p, int mode, bool count, hipStream_t s);
int pb_launch_ray_setup(const RenderParams& p, bool count, hipStream_t s);
int pb_launch_bin_count(const RenderParams& p, bool count, hipStream_t s);
int pb_fused_queue_macros(void);
bool pb_pipelined_eval(const RenderParams& p, int mode);
// Lane-per-macro-tile binning (pb_render.hip): segments per wave and blocks of PB_MBIN_WAVES waves for a segment count.
constexpr int PB_MBIN_WAVES = 4;
PB_HDI int pb_mbin_spw(int n_seg) { return n_seg <= 4096 ? 4 : (n_seg <= 32768 ? 16 : 32); }
PB_HDI int pb_mbin_blocks(int n_seg) { const int per = PB_MBIN_WAVES * pb_mbin_spw(n_seg); return (n_seg + per - 1) / per; }
// Evaluation items a list of cnt candidates is cut into.
PB_HDI uint32_t pb_item_slices(const RenderParams& p, uint32_t cnt) {
  return cnt <= (uint32_t)p.slice_whole ? (cnt ? 1u : 0u) : (cnt + (uint32_t)p.slice - 1u) / (uint32_t)p.slice;
}
// The residue queues of a relaxed-arithmetic heterogeneous frame (RenderParams::residue): items append to queue
// (item index mod PB_RESIDUE_QUEUES), so that their atomics spread over as many addresses (one address takes about
// 12 ns per atomic on this chip).  Layout in uint2 entries: PB_RESIDUE_HEAD entries of counters (queue q's at
// 32-bit word q * 32, 128 bytes apart), then the queues, pb_residue_queue_capacity() entries each.
constexpr int PB_RESIDUE_QUEUES = 256;
constexpr int PB_RESIDUE_HEAD = PB_RESIDUE_QUEUES * 32 / 2;
PB_HDI uint32_t pb_residue_queue_capacity(int total_entries) { return (uint32_t)(total_entries - PB_RESIDUE_HEAD) / (uint32_t)PB_RESIDUE_QUEUES; }
PB_HDI int pb_macro_count(const RenderParams& p) {
  return ((p.tiles_x + p.macro_w - 1) / p.macro_w) * ((p.tiles_y + p.macro_h - 1) / p.macro_h);
}
int pb_launch_bin_fill_eval(const RenderParams& p, int mode, bool count, hipStream_t s);
// stage: 0 = the whole frame, 1 = only what comes before the evaluation, 2 = only the evaluation and what follows it
int pb_launch_fast_frame(const RenderParams& p, int mode, bool lists_in_place, hipStream_t s, int stage = 0);
// One pass over the item queue that tests every item's candidates against its tile's pyramid: write = false counts the
// survivors (surv_cnt[item]); write = true stores them at surv_list + surv_cnt[n_items + item] and repoints the item.
int pb_launch_refine_items(const RenderParams& p, bool write, hipStream_t s);
int pb_max_item_survivors(void);      // the shortest LDS survivor list among the kernels that evaluate items
// The kept queue cut anew into items of at most `length` survivors (regroup_items_kernel).
int pb_launch_regroup_items(const RenderParams& p, bool write, int n_old, int length, uint32_t* regroup, uint32_t* items_kept, hipStream_t s);
bool pb_residue_frame(const RenderParams& p, int mode);   // the frame's evaluation hands pairs to a residue launch (pixels written after it)

#endif
