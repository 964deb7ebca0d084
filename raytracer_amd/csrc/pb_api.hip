// extern "C" layer of libphoton_beams.so (declarations and reference citations:
// include/photon_beams.h).  Host code here only moves data, launches kernels and
// keeps the context; nothing of the radiance path runs on the CPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>
#include <map>

#include "pb_internal.h"
#include "pb_scan.h"
#include "pb_grid.h"

static thread_local std::string g_last_error;
void pb_set_error(const std::string& msg) { g_last_error = msg; }

namespace {

int ensure_device(pb_ctx* ctx) {
  PB_HIP(hipSetDevice(ctx->device));
  return PB_OK;
}

// Candidate planes / quadrant flags of HitBoundingBox for the origin (0,0,0)
// (Intersections.h:95-112); same rule as the device-side gate_setup.
void host_gate_setup(const float lo[3], const float hi[3], GateBox& g) {
  uint32_t q = 0;
  bool inside = true;
  for (int i = 0; i < 3; ++i) {
    g.lo[i] = lo[i]; g.hi[i] = hi[i];
    if (0.0f < lo[i]) { g.cand[i] = lo[i]; inside = false; }
    else if (0.0f > hi[i]) { g.cand[i] = hi[i]; inside = false; }
    else { g.cand[i] = 0.0f; q |= (1u << i); }
  }
  if (inside) q |= 8u;
  g.qflags = q;
}


// ---------------------------------------------------------------------------------------------------------
// The workspace of pb_render_noalloc_d: one file-scope device array per GPU (every GPU the library runs on holds
// its own copy at its own address until the process ends), laid out once per device by static_workspace_init.
// In host code `scratch` is only a handle for hipGetSymbolAddress; kernels and memsets get the address it returns.
// ---------------------------------------------------------------------------------------------------------
constexpr size_t NOALLOC_SLOTS = PB_NOALLOC_MAX_QUEUE_SLOTS;
static_assert(NOALLOC_SLOTS == 2 * (size_t)PB_NOALLOC_MAX_GROUPS, "the static workspace is laid out for two slots per ray group");     // queue slots (macro tiles are padded to 64 tiles)
constexpr size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t SCRATCH_CTRL = 0;
constexpr size_t SCRATCH_RAYS = SCRATCH_CTRL + 256;
constexpr size_t SCRATCH_GROUP_LAST = SCRATCH_RAYS + align256(sizeof(float) * PB_RAY_WORDS * 64 * (size_t)PB_NOALLOC_MAX_GROUPS);
constexpr size_t SCRATCH_TILE_U32 = SCRATCH_GROUP_LAST + align256(sizeof(int) * (size_t)PB_NOALLOC_MAX_GROUPS);
constexpr size_t SCRATCH_SCAN_TMP = SCRATCH_TILE_U32 + align256(sizeof(uint32_t) * (4 + 3 * (size_t)PB_BIN_SUBS) * NOALLOC_SLOTS);
constexpr size_t SCRATCH_BIN_LIST = SCRATCH_SCAN_TMP + align256(sizeof(uint32_t) * (NOALLOC_SLOTS * PB_BIN_SUBS / 256 + 64));
constexpr size_t NOALLOC_BLK_WORDS = (size_t)1 << 18;                  // (binning block, macro tile) counters: 1 MiB
constexpr size_t SCRATCH_BLK = SCRATCH_BIN_LIST + align256(sizeof(uint32_t) * (size_t)PB_NOALLOC_MAX_LIST_ENTRIES);
constexpr size_t SCRATCH_ITEMS = SCRATCH_BLK + align256(sizeof(uint32_t) * NOALLOC_BLK_WORDS);
constexpr size_t SCRATCH_ACCUM = SCRATCH_ITEMS + align256(sizeof(uint32_t) * 4 * (size_t)PB_NOALLOC_MAX_ITEMS);
constexpr size_t SCRATCH_BYTES = SCRATCH_ACCUM + align256(sizeof(unsigned long long) * PB_ACC_PLANES * 64 * (size_t)PB_NOALLOC_MAX_GROUPS);
static_assert(SCRATCH_BYTES <= (size_t)PB_NOALLOC_SCRATCH_BYTES, "the layout must fit the stated size of the static workspace");

static __device__ __align__(256) unsigned char scratch[PB_NOALLOC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
struct StaticWorkspace { unsigned char* base = nullptr; FrameWorkspace ws; };
StaticWorkspace g_static[MAX_DEVICES];
std::mutex g_static_mu;

// With `device` current: the address of this GPU's copy of `scratch`, and the fixed workspace laid out in it.
int static_workspace_init(int device) {
  if (device < 0 || device >= MAX_DEVICES) { pb_set_error("pb_create: device index beyond the static workspace table"); return PB_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_static_mu);
  StaticWorkspace& sw = g_static[device];
  if (sw.base) return PB_OK;
  void* ptr = nullptr;
  PB_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) { pb_set_error("pb_create: hipGetSymbolAddress returned no address for the static workspace"); return PB_ERR_DEVICE; }
  unsigned char* b = (unsigned char*)ptr;
  FrameWorkspace& ws = sw.ws;
  ws.fixed = true;
  ws.d_ctrl = (uint32_t*)(b + SCRATCH_CTRL);
  ws.d_rays = (float*)(b + SCRATCH_RAYS);
  ws.d_group_last = (int*)(b + SCRATCH_GROUP_LAST);
  ws.ws_groups = PB_NOALLOC_MAX_GROUPS;
  ws.ws_chunks = 0;                       // the ordered cull modes are not served from the static workspace
  ws.d_tile_u32 = (uint32_t*)(b + SCRATCH_TILE_U32);
  ws.d_scan_tmp = (uint32_t*)(b + SCRATCH_SCAN_TMP);
  ws.ws_tiles = (int)NOALLOC_SLOTS;
  ws.d_bin_list = (uint32_t*)(b + SCRATCH_BIN_LIST);
  ws.ws_bin_cap = PB_NOALLOC_MAX_LIST_ENTRIES;
  ws.d_blk = (uint32_t*)(b + SCRATCH_BLK);
  ws.ws_blk_cap = NOALLOC_BLK_WORDS;
  ws.d_items = (uint32_t*)(b + SCRATCH_ITEMS);
  ws.ws_item_cap = PB_NOALLOC_MAX_ITEMS;
  ws.d_accum = (unsigned long long*)(b + SCRATCH_ACCUM);
  ws.ws_accum_px = (size_t)64 * PB_NOALLOC_MAX_GROUPS;
  sw.base = b;
  return PB_OK;
}

}  // namespace

namespace {
PbTune g_tune;
std::once_flag g_tune_once;
int env_int(const char* name, int unset) { const char* e = std::getenv(name); return e ? std::atoi(e) : unset; }
}  // namespace

void pb_tune_reload() {
  PbTune t;
  t.surv = env_int("PB_TUNE_SURV", -1); t.fused_max = env_int("PB_TUNE_FUSED_MAX", -1);
  t.macrow = env_int("PB_TUNE_MACROW", -1); t.macroh = env_int("PB_TUNE_MACROH", -1);
  t.xcdrun = env_int("PB_TUNE_XCDRUN", -1); t.whole = env_int("PB_TUNE_WHOLE", -1);
  t.slice = env_int("PB_TUNE_SLICE", -1); t.waves = env_int("PB_TUNE_WAVES", -1); t.raywaves = env_int("PB_TUNE_RAYWAVES", -1);
#ifdef PB_EXPERIMENTS
  // timing experiments that leave parts of a frame out (wrong frames): only in a library built with -DPB_EXPERIMENTS
  // (make HIPFLAGS+=-DPB_EXPERIMENTS), never in the one that ships
  t.debug = env_int("PB_DEBUG_FLAGS", 0);
#endif
  if (const char* e = std::getenv("PB_TUNE_INLINE")) t.inline_ratio = std::atof(e);
  t.no_prefine = std::getenv("PB_TUNE_NO_PREFINE") != nullptr; t.no_keep = std::getenv("PB_TUNE_NO_KEEP") != nullptr;
  t.separate_finalize = std::getenv("PB_SEPARATE_FINALIZE") != nullptr; t.refill = std::getenv("PB_TUNE_REFILL") != nullptr;
  t.debug_times = std::getenv("PB_DEBUG_TIMES") != nullptr; t.emit_twice = std::getenv("PB_TUNE_EMIT_TWICE") != nullptr;
  t.het1 = std::getenv("PB_TUNE_HET1") != nullptr; t.het2 = std::getenv("PB_TUNE_HET2") != nullptr;
  t.old_eval = std::getenv("PB_OLD_EVAL") != nullptr; t.no_residue = std::getenv("PB_TUNE_NO_RESIDUE") != nullptr;
  t.wide = std::getenv("PB_TUNE_WIDE") != nullptr; t.het_lds_list = std::getenv("PB_TUNE_HET_LDS_LIST") != nullptr;
  t.het1_lds_tab = std::getenv("PB_TUNE_HET1_LDS_TAB") != nullptr;
  g_tune = t;
}

const PbTune& pb_tune() {
  std::call_once(g_tune_once, pb_tune_reload);
  return g_tune;
}

int pb_ensure_segment_capacity(pb_ctx* ctx, int n) {
  if (n <= ctx->cap_seg) return PB_OK;
  int cap = n + n / 8 + 64;
  if (ctx->d_segs) PB_HIP(hipFree(ctx->d_segs));
  if (ctx->d_frames) PB_HIP(hipFree(ctx->d_frames));
  if (ctx->d_capA) PB_HIP(hipFree(ctx->d_capA));
  if (ctx->d_capB) PB_HIP(hipFree(ctx->d_capB));
  if (ctx->d_het) PB_HIP(hipFree(ctx->d_het));
  if (ctx->d_het_axis) PB_HIP(hipFree(ctx->d_het_axis));
  if (ctx->d_het_fast) PB_HIP(hipFree(ctx->d_het_fast));
  ctx->d_het = nullptr; ctx->d_het_axis = nullptr; ctx->d_het_fast = nullptr;
  ctx->d_segs = nullptr; ctx->d_frames = nullptr; ctx->d_capA = nullptr; ctx->d_capB = nullptr; ctx->cap_seg = 0;
  PB_HIP(hipMalloc(&ctx->d_segs, sizeof(pb_segment) * (size_t)cap));
  PB_HIP(hipMalloc(&ctx->d_frames, sizeof(SegFrame) * (size_t)cap));
  PB_HIP(hipMalloc(&ctx->d_capA, sizeof(float4) * (size_t)cap));
  PB_HIP(hipMalloc(&ctx->d_capB, sizeof(float4) * (size_t)cap));
  PB_HIP(hipMalloc(&ctx->d_het, sizeof(HetFrame) * (size_t)cap));
  PB_HIP(hipMalloc(&ctx->d_het_axis, sizeof(BeamAxis) * (size_t)cap));
  PB_HIP(hipMalloc(&ctx->d_het_fast, sizeof(HetFast) * (size_t)cap));
  ctx->cap_seg = cap;
  return PB_OK;
}

namespace {

int default_mode(const pb_ctx* ctx) {
  if (ctx->ctl.adaptive) return PB_MODE_GENERIC;      // cones exist only in BeamRadiance (BeamRadiance.h:488)
  if (ctx->ctl.heterogeneous && ctx->ctl.one_dimensional) return PB_MODE_73HET;
  return ctx->ctl.one_dimensional ? PB_MODE_73 : PB_MODE_721;
}

}  // namespace

void pb_note_reader(pb_ctx* ctx, hipStream_t s) {
  for (int i = 0; i < ctx->n_reader_streams; ++i) if (ctx->reader_streams[i] == s) return;
  if (ctx->n_reader_streams < 8) ctx->reader_streams[ctx->n_reader_streams++] = s;
  else ctx->readers_overflow = true;
}

int pb_wait_for_readers(pb_ctx* ctx, hipStream_t writer, bool host_waits) {
  bool whole_device = ctx->readers_overflow;
  for (int i = 0; i < ctx->n_reader_streams && !whole_device; ++i) {
    hipStream_t r = ctx->reader_streams[i];
    if (r == writer) continue;          // stream order does it
    // a stream the caller has destroyed since: its work has been waited for or abandoned; the device-wide wait below
    // covers either case
    if (hipEventRecord(ctx->ev_readers, r) != hipSuccess) { (void)hipGetLastError(); whole_device = true; break; }
    PB_HIP(hipStreamWaitEvent(writer, ctx->ev_readers, 0));
  }
  if (whole_device) PB_HIP(hipDeviceSynchronize());
  if (whole_device || host_waits) {
    // nothing noted is still of interest: either it has finished, or the caller's host wait on `writer` covers it
    const bool keep_writer = !whole_device;
    ctx->n_reader_streams = 0; ctx->readers_overflow = false;
    if (keep_writer) {
      // frames on the writer's own stream are ordered with this call but not with a later writer on another stream
      ctx->reader_streams[ctx->n_reader_streams++] = writer;
    }
  }
  return PB_OK;
}

int pb_prepare_frames(pb_ctx* ctx, int mode, hipStream_t s) {
  ctx->prepare_stream = s;
  PB_HIP(hipEventRecord(ctx->ev[2], s));
  int rc = pb_launch_prepare_segments(ctx->d_segs, ctx->n_seg, mode, ctx->ctl.one_dimensional, ctx->ctl.scattering_c,
                                      ctx->ctl.absorption_c + ctx->ctl.scattering_c, ctx->d_frames, ctx->d_capA, ctx->d_capB, ctx->d_het,
                                      ctx->d_het_fast, s);
  if (rc != PB_OK) return rc;
  if (mode != PB_MODE_721) {     // the estimators that go through Integral_73 / Integral_73Het
    rc = pb_launch_beam_axis(ctx->d_capA, ctx->d_capB, ctx->d_het, ctx->n_seg, ctx->d_het_axis, s);
    if (rc != PB_OK) return rc;
  }
  PB_HIP(hipEventRecord(ctx->ev[3], s));
  ctx->ev_valid[1] = true;
  ctx->frame_mode = mode;
  return PB_OK;
}

int pb_default_mode(const pb_ctx* ctx) { return default_mode(ctx); }

namespace {

// A need beyond a fixed workspace is an error of the caller's frame, named so that it can be told what to shrink.
int beyond_fixed(const char* what, size_t need, size_t cap) {
  pb_set_error(std::string("pb_render_noalloc_d: the frame needs ") + std::to_string(need) + " " + what +
               ", the static workspace holds " + std::to_string(cap) + " (see PB_NOALLOC_* in photon_beams.h)");
  return PB_ERR_CAPACITY;
}

int ensure_workspace(FrameWorkspace& ws, int groups, int chunks) {
  if (ws.fixed) {
    if (groups > ws.ws_groups) return beyond_fixed("ray groups (8x8-pixel tiles x sub-samples)", (size_t)groups, (size_t)ws.ws_groups);
    if (chunks > ws.ws_chunks) return beyond_fixed("candidate chunks", (size_t)chunks, (size_t)ws.ws_chunks);
    return PB_OK;
  }
  if (groups > ws.ws_groups) {
    if (ws.d_rays) PB_HIP(hipFree(ws.d_rays));
    if (ws.d_group_last) PB_HIP(hipFree(ws.d_group_last));
    ws.d_rays = nullptr; ws.d_group_last = nullptr; ws.ws_groups = 0;
    PB_HIP(hipMalloc(&ws.d_rays, sizeof(float) * PB_RAY_WORDS * 64 * (size_t)groups));
    PB_HIP(hipMalloc(&ws.d_group_last, sizeof(int) * (size_t)groups));
    ws.ws_groups = groups;
  }
  if (chunks > ws.ws_chunks) {
    if (ws.d_chunks) PB_HIP(hipFree(ws.d_chunks));
    if (ws.d_partials) PB_HIP(hipFree(ws.d_partials));
    ws.d_chunks = nullptr; ws.d_partials = nullptr; ws.ws_chunks = 0;
    PB_HIP(hipMalloc(&ws.d_chunks, sizeof(uint32_t) * PB_CHUNK_WORDS * (size_t)chunks));
    PB_HIP(hipMalloc(&ws.d_partials, sizeof(double) * 3 * 64 * (size_t)chunks));
    ws.ws_chunks = chunks;
  }
  return PB_OK;
}

int ensure_bin_workspace(FrameWorkspace& ws, int tiles, int bin_cap, int item_cap, size_t accum_px) {
  if (ws.fixed) {
    if (tiles > ws.ws_tiles) return beyond_fixed("queue slots (64 per 64x64-pixel macro tile)", (size_t)tiles, (size_t)ws.ws_tiles);
    if (bin_cap > ws.ws_bin_cap) return beyond_fixed("macro-tile list entries", (size_t)bin_cap, (size_t)ws.ws_bin_cap);
    if (item_cap > ws.ws_item_cap) return beyond_fixed("evaluation items", (size_t)item_cap, (size_t)ws.ws_item_cap);
    if (accum_px > ws.ws_accum_px) return beyond_fixed("pixel sums", accum_px, ws.ws_accum_px);
    return PB_OK;
  }
  if (tiles > ws.ws_tiles) {
    if (ws.d_tile_u32) PB_HIP(hipFree(ws.d_tile_u32));
    if (ws.d_scan_tmp) PB_HIP(hipFree(ws.d_scan_tmp));
    ws.d_tile_u32 = nullptr; ws.d_scan_tmp = nullptr; ws.ws_tiles = 0;
    PB_HIP(hipMalloc(&ws.d_tile_u32, sizeof(uint32_t) * (4 + 3 * (size_t)PB_BIN_SUBS) * (size_t)tiles));   // macro tiles <= tiles
    PB_HIP(hipMalloc(&ws.d_scan_tmp, sizeof(uint32_t) * pb_scan_tmp_words(tiles * PB_BIN_SUBS)));
    ws.ws_tiles = tiles;
  }
  if (bin_cap > ws.ws_bin_cap) {
    if (ws.d_bin_list) PB_HIP(hipFree(ws.d_bin_list));
    ws.d_bin_list = nullptr; ws.ws_bin_cap = 0;
    PB_HIP(hipMalloc(&ws.d_bin_list, sizeof(uint32_t) * (size_t)bin_cap));
    ws.ws_bin_cap = bin_cap;
  }
  if (item_cap > ws.ws_item_cap) {
    if (ws.d_items) PB_HIP(hipFree(ws.d_items));
    ws.d_items = nullptr; ws.ws_item_cap = 0;
    PB_HIP(hipMalloc(&ws.d_items, sizeof(uint32_t) * 4 * (size_t)item_cap));
    ws.ws_item_cap = item_cap;
  }
  if (accum_px > ws.ws_accum_px) {
    if (ws.d_accum) PB_HIP(hipFree(ws.d_accum));
    ws.d_accum = nullptr; ws.ws_accum_px = 0;
    PB_HIP(hipMalloc(&ws.d_accum, sizeof(unsigned long long) * PB_ACC_PLANES * accum_px));
    ws.ws_accum_px = accum_px;
  }
  return PB_OK;
}

int ensure_blk(FrameWorkspace& ws, size_t need) {
  if (need <= ws.ws_blk_cap) return PB_OK;
  if (ws.fixed) return beyond_fixed("(binning block, macro tile) counters", need, ws.ws_blk_cap);
  if (ws.d_blk) PB_HIP(hipFree(ws.d_blk));
  ws.d_blk = nullptr; ws.ws_blk_cap = 0;
  PB_HIP(hipMalloc(&ws.d_blk, sizeof(uint32_t) * need));
  ws.ws_blk_cap = need;
  return PB_OK;
}

// The kept queue of a (segment set, frame geometry): every item's candidates are tested against its tile's pyramid once
// (refine_items_kernel: count, prefix sum, write) and the survivors, one contiguous run per tile, are cut into items of
// equal length (regroup_items_kernel).  Two read-backs (survivors in all, items in all): this runs once per segment set
// and geometry, in a frame that has waited for the device already.  The rays' tile planes and group_last of this
// frame geometry must be in the workspace.  p.items / p.tile_items still describe the queue as it was built.
int build_survivor_lists(pb_ctx* ctx, FrameWorkspace& ws, RenderParams& p, int mode, int slots, hipStream_t s) {
  ws.surv_ready = false;
  const int n_old = p.n_items_host;
  if (n_old <= 0 || ws.fixed) return PB_OK;
  const size_t scan_words = pb_scan_tmp_words(n_old > slots ? n_old : slots);
  const size_t words = 2 * (size_t)n_old + 1 + 2 * (size_t)slots + scan_words + 8;
  if (words > ws.ws_surv_words) {
    if (ws.d_surv_cnt) PB_HIP(hipFree(ws.d_surv_cnt));
    ws.d_surv_cnt = nullptr; ws.ws_surv_words = 0;
    const size_t cap = words + words / 4;
    PB_HIP(hipMalloc(&ws.d_surv_cnt, sizeof(uint32_t) * cap));
    ws.ws_surv_words = cap;
  }
  p.surv_cnt = ws.d_surv_cnt;
  uint32_t* off = ws.d_surv_cnt + n_old;                 // n_old + 1 entries
  uint32_t* regroup_cnt = off + n_old + 1;
  uint32_t* regroup_off = regroup_cnt + slots;
  uint32_t* scan_tmp = regroup_off + slots;
  uint32_t* d_items_total = scan_tmp + scan_words;
  int rc = pb_launch_refine_items(p, false, s);
  if (rc != PB_OK) return rc;
  rc = pb_exclusive_scan_u32(p.surv_cnt, off, n_old, off + n_old, scan_tmp, s);
  if (rc != PB_OK) return rc;
  PB_HIP(hipMemcpyAsync(ctx->h_ctrl, off + n_old, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  const size_t total = ctx->h_ctrl[0];
  // Item length.  A frame with little work in all is bound by its longest item (a wave walks its survivors one after
  // the other while the rest of the chip idles), a frame with much work by what every item pays before and after its
  // pair loop (rays, sums): about three items per wave the chip can hold, between 12 and 64 survivors each.
  // Measured (survivors per item: ms a frame): 512^2 / 1000 photons 12: 0.0345, 16: 0.0335, 24: 0.0336, 32: 0.0368;
  // 10 000 photons 32: 0.162, 48: 0.159, 64: 0.163, 96: 0.173; 1024^2 / 100 000 photons 64: 3.13, 128: 2.94, 192: 2.92,
  // 256: 2.93; the heterogeneous estimator (its items end with a partly filled drain of queued pairs, but a pair costs
  // it four times as much) 32: 0.728, 48: 0.720, 64: 0.716, 96: 0.733, 128: 0.735.
  const bool queued = mode != PB_MODE_721;
  int length = (int)(total / ((queued ? 6 : 3) * 8192));
  const int longest = queued ? 128 : 192;
  // (a frame short enough to be one launch, its items setting up their own rays: 16: 0.0307, 24: 0.0289, 32: 0.0317)
  const int shortest = queued ? 16 : 24;
  length = length < shortest ? shortest : (length > longest ? longest : length);
  if (p.strip_stride > 1) length = length / 2 < 16 ? 16 : length / 2;       // a rank's share of the strips: fewer items
  // an item's survivors must fit the LDS list of whichever kernel evaluates it
  const int most = pb_max_item_survivors();
  if (pb_tune().surv >= 1) length = pb_tune().surv;
  if (length > most) length = most;
  PB_HIP(hipMemsetAsync(regroup_cnt, 0, sizeof(uint32_t) * (size_t)slots, s));
  rc = pb_launch_regroup_items(p, false, n_old, length, regroup_cnt, nullptr, s);
  if (rc != PB_OK) return rc;
  rc = pb_exclusive_scan_u32(regroup_cnt, regroup_off, slots, d_items_total, scan_tmp, s);
  if (rc != PB_OK) return rc;
  PB_HIP(hipMemcpyAsync(ctx->h_ctrl, d_items_total, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  const size_t n_new = ctx->h_ctrl[0];
  if (n_new > (size_t)(1u << 30)) return PB_OK;          // not kept: the frame goes on with the queue as built
  if (total > ws.ws_surv_cap) {
    if (ws.d_surv_list) PB_HIP(hipFree(ws.d_surv_list));
    ws.d_surv_list = nullptr; ws.ws_surv_cap = 0;
    const size_t cap = total + total / 8 + 1024;
    PB_HIP(hipMalloc(&ws.d_surv_list, sizeof(uint32_t) * cap));
    ws.ws_surv_cap = cap;
  }
  if (n_new > (size_t)ws.ws_items_kept_cap) {
    if (ws.d_items_kept) PB_HIP(hipFree(ws.d_items_kept));
    ws.d_items_kept = nullptr; ws.ws_items_kept_cap = 0;
    const size_t cap = n_new + n_new / 8 + 1024;
    PB_HIP(hipMalloc(&ws.d_items_kept, sizeof(uint32_t) * 4 * cap));
    ws.ws_items_kept_cap = (int)cap;
  }
  p.surv_list = ws.d_surv_list;
  rc = pb_launch_refine_items(p, true, s);
  if (rc != PB_OK) return rc;
  rc = pb_launch_regroup_items(p, true, n_old, length, regroup_off, ws.d_items_kept, s);
  if (rc != PB_OK) return rc;
  ws.ws_items_kept = (int)n_new;
  ws.surv_ready = true;
  return PB_OK;
}

// Frame through screen-space bins (cull mode 2); p is complete except for the workspace pointers.
int render_binned(pb_ctx* ctx, FrameWorkspace& ws, RenderParams& p, int mode, const unsigned long long key_in[6], hipStream_t s) {
  // Lane-per-macro-tile binning tests every segment against every macro tile and keeps a counter per (block of 128
  // segments, macro tile), which one block sums: the right trade for the segment sets a frame usually has, not
  // for millions of segments (23.7 M segments, a 4096 x 128 band: 85 ms a frame against 11 ms for the walk over
  // each segment's own rectangle of macro tiles).  The two ways fill the lists by different (conservative) tests, so
  // what was sized for one says nothing about the other: the limit is part of the key.
  int fused_max_seg = 1 << 20;
  if (pb_tune().fused_max >= 0) fused_max_seg = pb_tune().fused_max;
  unsigned long long key[6];
  std::memcpy(key, key_in, sizeof(key));
  key[4] |= ((unsigned long long)(uint32_t)fused_max_seg + 1ull) << 8;
  // whether the kept queue's items hold survivors (build_survivor_lists) or candidates is part of what is kept
  const bool keep_survivors = !ws.fixed && !pb_tune().no_prefine;
  key[4] |= (keep_survivors ? 1ull : 0ull) << 48;
  const int tiles = p.tiles_x * p.tiles_y;
  const int groups = tiles * p.ssaa * p.ssaa;
  const size_t px = (size_t)tiles * 64;      // pixel sums are stored per tile
  bool verified = std::memcmp(key, ws.ws_key, sizeof(unsigned long long) * 6) == 0;
  // The item length chosen for a frame geometry is tried first for the next segment set rendered with that geometry (a
  // stop-motion loop re-casts every frame, and choosing anew costs a second counting pass); it is chosen again where
  // the new set's lists call for a length more than a quarter away.
  const bool same_geometry = std::memcmp(key + 1, ws.ws_key + 1, sizeof(unsigned long long) * 5) == 0;
  if (!verified) {
    // From here on this frame's passes overwrite the workspace, and they can still fail part-way (a fixed workspace
    // learns that a frame does not fit only after the counting pass has run): what the workspace held for the old
    // key is gone either way, so the key goes first and comes back only where a frame has been sized completely.
    ws.queue_kept = false;
    std::memset(ws.ws_key, 0, sizeof(ws.ws_key));
    ws.ws_items_needed = 0;
  }
  bool slice_settled = verified;
  if (ctx->tune_slice <= 0) {
    if ((verified || same_geometry) && ws.ws_slice > 0) p.slice = ws.ws_slice;
    else if (!verified) ws.ws_slice = 0;
  }
  int want_bins = ws.ws_bin_cap, want_items = ws.ws_item_cap;
  if (!verified) {
    if (want_bins < tiles * 64 + 1024) want_bins = tiles * 64 + 1024;
    if (want_items < groups * 2 + 1024) want_items = groups * 2 + 1024;
  }
  for (int attempt = 0; attempt < 4; ++attempt) {
    int rc = ensure_workspace(ws, groups, 0);
    if (rc != PB_OK) return rc;
    // With interleaved strips a macro tile takes fewer tile rows: eight would span strip_stride times the
    // height on screen and hand every tile the segments of a whole column to refine.
    p.macro_w = PB_MACRO;
    if (pb_tune().macrow >= 1 && pb_tune().macrow <= PB_MACRO) p.macro_w = pb_tune().macrow;
    p.macro_h = p.macro_w / p.strip_stride;     // keeps a macro tile's footprint square on screen
    if (p.macro_h < 1) p.macro_h = 1;
    if (pb_tune().macroh >= 1 && pb_tune().macroh <= PB_MACRO) p.macro_h = pb_tune().macroh;
    const int mtiles = pb_macro_count(p);
    const int slots = mtiles * p.macro_w * p.macro_h;     // queue slots: every macro tile has room for all its tiles
    rc = ensure_bin_workspace(ws, slots, want_bins, want_items, px);
    if (rc != PB_OK) return rc;
    p.rays = ws.d_rays; p.group_last = ws.d_group_last; p.ctrl = ws.d_ctrl;
    p.walk = nullptr;
    if (mode == PB_MODE_73HET && p.grid_z && !p.precise && !ws.fixed && p.grid_side <= VIEW_WALK_MAX_SIDE) {
      // the relaxed-arithmetic estimator walks every view ray through the grid once per frame
      if (groups > ws.ws_walk_groups) {
        if (ws.d_walk) PB_HIP(hipFree(ws.d_walk));
        ws.d_walk = nullptr; ws.ws_walk_groups = 0;
        PB_HIP(hipMalloc(&ws.d_walk, sizeof(float) * VIEW_WALK_STRIDE * 64 * (size_t)groups));
        ws.ws_walk_groups = groups;
      }
      p.walk = ws.d_walk;
    }
    p.residue = nullptr; p.residue_capacity = 0;
    if ((mode == PB_MODE_73HET || mode == PB_MODE_73) && !p.precise && !ws.fixed) {
      // pairs the relaxed estimator cannot decide wait for a launch of their own, so that no item ends with a
      // nearly empty wave running the exact-order code; an item whose pairs do not fit evaluates them itself
      const long long wanted = (long long)groups * 512;
      const int want = wanted < (1 << 18) ? (1 << 18) : (wanted > (1 << 28) ? (1 << 28) : (int)wanted);
      if (want > ws.ws_residue_cap) {
        if (ws.d_residue) PB_HIP(hipFree(ws.d_residue));
        ws.d_residue = nullptr; ws.ws_residue_cap = 0;
        PB_HIP(hipMalloc(&ws.d_residue, sizeof(uint2) * (size_t)want));
        ws.ws_residue_cap = want;
      }
      p.residue = ws.d_residue; p.residue_capacity = ws.ws_residue_cap;
      if (!pb_residue_frame(p, mode)) { p.residue = nullptr; p.residue_capacity = 0; }   // another kernel evaluates this frame
    }
    {
      const size_t T = (size_t)ws.ws_tiles;
      p.tile_count = ws.d_tile_u32; p.item_off = ws.d_tile_u32 + T;
      p.macro_cnt = ws.d_tile_u32 + 2 * T; p.macro_rank = ws.d_tile_u32 + 3 * T;
      p.sub_count = ws.d_tile_u32 + 4 * T; p.sub_fill = p.sub_count + T * PB_BIN_SUBS; p.sub_off = p.sub_fill + T * PB_BIN_SUBS;
    }
    p.bin_list = ws.d_bin_list; p.items = ws.d_items; p.accum = ws.d_accum;
    p.tile_items = p.tile_done = nullptr; p.finalize_in_eval = 0;
    p.bin_capacity = ws.ws_bin_cap; p.item_capacity = ws.ws_item_cap;
    p.fused_queue = (mtiles <= pb_fused_queue_macros() && p.n_seg <= fused_max_seg) ? 1 : 0;
    // frames whose queue one block builds bin without atomics: the counters' space holds the per-tile item counts
    if (p.fused_queue) { p.tile_items = p.sub_count; p.tile_done = p.sub_fill; }
    p.bin_mask = nullptr;
    p.blk = nullptr;
    if (p.fused_queue && p.n_seg > 0) {
      rc = ensure_blk(ws, (size_t)pb_mbin_blocks(p.n_seg) * (size_t)mtiles);
      if (rc != PB_OK) return rc;
      p.blk = ws.d_blk;
    }
    if (ctx->count_on || ctx->time_render) PB_HIP(hipEventRecord(ctx->ev[0], s));
    if (verified && ws.queue_kept && p.fused_queue && !ctx->count_on && p.n_seg > 0 && !pb_tune().no_keep) {
      // The lists, their offsets and the item queue are functions of the segment set and the frame geometry alone and are
      // in the workspace from the frame that sized it for this key (as a tree would be kept): ray set-up, evaluation,
      // pixels.  PB_TUNE_REFILL fills the lists again every frame (the binning pass beside the ray set-up).
      p.n_items_host = ws.ws_items_needed;
      p.finalize_in_eval = (pb_pipelined_eval(p, mode) && !pb_residue_frame(p, mode) && !pb_tune().separate_finalize) ? 1 : 0;
      bool made_now = false;
      if (!ws.surv_ready && keep_survivors) {
        // the first kept frame of this queue: it also tests every item's candidates once and for all and cuts the
        // queue anew by survivors; its ray set-up runs first (the tests read the tiles' planes from the ray blocks and
        // group_last), against the per-tile item counts of the queue as built
        rc = pb_launch_fast_frame(p, mode, !pb_tune().refill, s, 1);
        if (rc != PB_OK) return rc;
        rc = build_survivor_lists(ctx, ws, p, mode, slots, s);
        if (rc != PB_OK) return rc;
        made_now = ws.surv_ready;
      }
      if (ws.surv_ready) {
        p.items = ws.d_items_kept; p.item_capacity = ws.ws_items_kept_cap; p.n_items_host = ws.ws_items_kept;
        p.surv_list = ws.d_surv_list; p.kept_queue = 1;
      }
      // runs of 64 queue items per XCD: a tile's items share an L2 (cfg2: 91 -> 60 MB fetched by the evaluation, same
      // time; 512^2 / 1000 photons 0.0447 -> 0.0420 ms; longer runs leave XCDs idle at the end: cfg2 0.154 ms at 64, 0.160
      // at 512, 0.170 at 1024).  A frame of very many items has rounds to spare: in runs of 4096 an XCD works through a
      // few macro tiles at a time (1024^2, 100 000 photons, 186 000 items: 4.9 -> 3.4 GB through the fabric, L2 misses
      // 3.2e7 -> 2.2e7 of 7.8e7 requests, 2.89 ms either way)
      if (p.xcd_run < 0) p.xcd_run = p.n_items_host <= (1 << 17) ? 64 : 4096;
      // A frame of few items per tile is one launch: every item sets up its own rays (eval_items_fast_kernel's INLINE
      // form), which costs a tile's ray set-up once per item but saves the launch before the evaluation and the rays'
      // trip through memory.  The sums must have been left clear by the frame before (its last arrivers do that).
      double inline_ratio = 4.0;
      if (pb_tune().inline_ratio >= 0.0) inline_ratio = pb_tune().inline_ratio;
      const bool one_launch = ws.surv_ready && !made_now && ws.sums_left_clear && mode == PB_MODE_721 && p.finalize_in_eval &&
                              pb_pipelined_eval(p, mode) && !ctx->ctl.short_view &&
                              (double)p.n_items_host <= inline_ratio * (double)groups;
      if (one_launch) {
        p.inline_item_blocks = 1;
        rc = pb_launch_fast_frame(p, mode, true, s, 2);
      }
      else if (made_now) {
        // the set-up above prepared the sums and the black tiles for the old per-tile item counts: once more for the new
        rc = pb_launch_fast_frame(p, mode, true, s, 1);
        if (rc != PB_OK) return rc;
        rc = pb_launch_fast_frame(p, mode, true, s, 2);
      }
      else if (keep_survivors && !ws.surv_ready) {       // the lists could not be made: stage 1 has run
        rc = pb_launch_fast_frame(p, mode, true, s, 2);
      }
      else rc = pb_launch_fast_frame(p, mode, !pb_tune().refill, s);
      if (rc != PB_OK) return rc;
      ws.sums_left_clear = mode == PB_MODE_721 && p.finalize_in_eval && pb_pipelined_eval(p, mode);
      if (ctx->count_on || ctx->time_render) { PB_HIP(hipEventRecord(ctx->ev[1], s)); ctx->ev_valid[0] = true; }
      return PB_OK;
    }
    ws.surv_ready = false;        // the passes below write the item queue anew, every item over its candidates again
    ws.sums_left_clear = false;
    if (!p.fused_queue) {
      PB_HIP(hipMemsetAsync(p.item_off, 0, sizeof(uint32_t) * (size_t)slots, s));   // slots of tiles outside the frame stay empty
      PB_HIP(hipMemsetAsync(ws.d_ctrl, 0, sizeof(uint32_t) * 4, s));
      PB_HIP(hipMemsetAsync(p.sub_count, 0, sizeof(uint32_t) * 2 * (size_t)ws.ws_tiles * PB_BIN_SUBS, s));   // sub_count, sub_fill
      PB_HIP(hipMemsetAsync(ws.d_accum, 0, sizeof(unsigned long long) * PB_ACC_PLANES * px, s));
    }
    rc = pb_launch_ray_setup(p, ctx->count_on, s);
    if (rc != PB_OK) return rc;
    rc = pb_launch_bin_count(p, ctx->count_on, s);     // count pass, then (small frames) the whole queue in one block
    if (rc != PB_OK) return rc;
    if (!p.fused_queue) {
      rc = pb_exclusive_scan_u32(p.sub_count, p.sub_off, mtiles * PB_BIN_SUBS, ws.d_ctrl + 2, ws.d_scan_tmp, s);
      if (rc != PB_OK) return rc;
      rc = pb_exclusive_scan_u32(p.item_off, p.item_off, slots, ws.d_ctrl + 0, ws.d_scan_tmp, s);
      if (rc != PB_OK) return rc;
    }
    if (verified) break;
    PB_HIP(hipMemcpyAsync(ctx->h_ctrl, ws.d_ctrl, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
    PB_HIP(hipStreamSynchronize(s));
    const uint32_t need_items = ctx->h_ctrl[0], need_bins = ctx->h_ctrl[2];
    if (ctx->tune_slice <= 0 && !slice_settled) {
      slice_settled = true;
      // Item length from the mean macro-tile list: a wave walks its item's survivors one L2 round trip after
      // the other, so a frame with short lists (few items, GPU far from full) wants short items, while long
      // lists amortise the per-item work better with longer ones.  Measured: 1052 segments 0.145 -> 0.100 ms at
      // 48 instead of 192, 17 424 segments best at 128.
      const uint32_t mean_list = need_bins / (uint32_t)(mtiles > 0 ? mtiles : 1);
      // The estimators that queue their hits (Integral_73Het, the generic dispatch) spend 14 times as long per
      // candidate and drain a partly filled queue at the end of every item: they want longer items (cfg3:
      // 3.37 ms at 48, 3.08 at 96, 2.98 at 128..192, 3.11 at 256; its mean list is about 300).
      const bool queued = mode != PB_MODE_721;
      // the relaxed-arithmetic kernels end an item with a partly filled drain of their queued pairs: longer items still
      // (cfg3, eval_items_het2_kernel with the residue launch: 1.30 ms at 96, 1.21 at 128, 1.18 at 256, 1.21 at 384,
      // 1.27 at 512; the one-pass kernel, which also copied its rays' walk tables: 2.09 / 1.57 / 1.47 / 1.38 / 1.42)
      const bool tabulated = queued && pb_pipelined_eval(p, mode);
      int want = (int)(mean_list / (tabulated ? 1 : (queued ? 2 : 6)) + 15) / 16 * 16;
      int longest = p.strip_stride >= 8 ? 48 : (p.strip_stride > 1 ? 96 : 128);   // a rank's share of strips has few items
      if (queued) longest = tabulated ? (pb_residue_frame(p, mode) ? longest * 2 : longest * 3) : longest * 3 / 2;
      // very long lists mean hundreds of thousands of items: the per-item work (rays, planes, atomics) is worth
      // amortising over more candidates (1024^2, 106 381 segments, mean list 6743: 3.62 ms at 128, 3.50 at 256,
      // 3.43 at 384, 3.45 at 512; 512^2, 17 424 segments, mean list 1574: 0.198 ms at 128, 0.206 at 192)
      else if (mean_list >= 4096) longest *= 3;
      const int shortest = queued ? 64 : 48;
      want = want < shortest ? shortest : (want > longest ? longest : want);
      if (ws.ws_slice > 0 && ws.ws_slice == p.slice && std::abs(want - p.slice) * 4 <= p.slice) want = p.slice;   // near enough: kept
      ws.ws_slice = want;
      if (want != p.slice) {                 // same capacities, queue rebuilt with the new length
        p.slice = want;
        if (ctx->count_on) PB_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * PB_CNT_N, s));
        attempt--;
        continue;
      }
    }
    if (need_items <= (uint32_t)ws.ws_item_cap && need_bins <= (uint32_t)ws.ws_bin_cap) {
      ws.ws_items_needed = (int)need_items;
      std::memcpy(ws.ws_key, key, sizeof(unsigned long long) * 6);
      verified = true;
      break;
    }
    if (need_items > (uint32_t)ws.ws_item_cap) want_items = (int)(need_items + need_items / 4 + 64);
    if (need_bins > (uint32_t)ws.ws_bin_cap) want_bins = (int)(need_bins + need_bins / 4 + 64);
    if (ctx->count_on) PB_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * PB_CNT_N, s));
  }
  if (!verified) { pb_set_error("pb_render: bin workspace could not be sized"); return PB_ERR_CAPACITY; }
  p.n_items_host = ws.ws_items_needed;
  if (p.xcd_run < 0) p.xcd_run = 0;
  int rc = pb_launch_bin_fill_eval(p, mode, ctx->count_on, s);
  if (rc != PB_OK) return rc;
  ws.queue_kept = p.fused_queue != 0 && p.n_seg > 0;
  if (ctx->count_on || ctx->time_render) { PB_HIP(hipEventRecord(ctx->ev[1], s)); ctx->ev_valid[0] = true; }
  return PB_OK;
}

int render_common(pb_ctx* ctx, FrameWorkspace& ws, int x0, int y0, int out_w, int out_h, int strip_stride, int mode, int use_lbvh,
                  float* d_out, hipStream_t s) {
  if (!ctx->d_tris) { pb_set_error("pb_render: pb_set_scene has not been called"); return PB_ERR_STATE; }
  if (!ctx->have_segments) { pb_set_error("pb_render: no segment set (pb_upload_segments / pb_emit_beams)"); return PB_ERR_STATE; }
  if (ws.fixed) {
    // shared by every context of this GPU and left as the last call left it: what was verified for another
    // context's frame says nothing about this one (nothing in it needs to be zero when a frame starts: whatever a
    // frame accumulates into it clears first)
    if (ws.owner != ctx->id) { std::memset(ws.ws_key, 0, sizeof(ws.ws_key)); ws.ws_slice = 0; ws.queue_kept = false; ws.owner = ctx->id; }
  }
  if (mode == PB_MODE_DEFAULT) mode = default_mode(ctx);
  if (mode != PB_MODE_721 && mode != PB_MODE_73HET && mode != PB_MODE_73 && mode != PB_MODE_GENERIC) {
    pb_set_error("pb_render: mode must be PB_MODE_721, PB_MODE_73, PB_MODE_73HET or PB_MODE_GENERIC"); return PB_ERR_ARG;
  }
  if (use_lbvh < 0 || use_lbvh > 2) { pb_set_error("pb_render: cull mode must be 0 (scan), 1 (LBVH) or 2 (screen bins)"); return PB_ERR_ARG; }
  if (use_lbvh == 1 && (ctx->lb_n < 0 || ctx->lb_version != ctx->segs_version)) {
    pb_set_error("pb_render: pb_build_lbvh has not been called for this segment set"); return PB_ERR_STATE;
  }
  if (ctx->frame_mode != mode) {
    // the records are rewritten for the other estimator: frames of the old one may still be running elsewhere
    int rc = pb_wait_for_readers(ctx, s, false);
    if (rc != PB_OK) return rc;
    rc = pb_prepare_frames(ctx, mode, s);
    if (rc != PB_OK) return rc;
  } else if (ctx->ev_valid[1] && ctx->prepare_stream != s) {
    // records prepared by a call queued on another stream and not waited for (pb_upload_segments_d)
    PB_HIP(hipStreamWaitEvent(s, ctx->ev[3], 0));
  }
  pb_note_reader(ctx, s);
  RenderParams p;
  std::memset(&p, 0, sizeof(p));
  p.tris = ctx->d_tris; p.n_tri = ctx->n_tri; p.tri_box = (const float4*)(ctx->d_tris + ctx->n_tri);
  p.tri_edge = (const float*)(p.tri_box + ctx->n_tri);
  p.frames = ctx->d_frames; p.capA = ctx->d_capA; p.capB = ctx->d_capB; p.n_seg = ctx->n_seg;
  host_gate_setup(ctx->root_lo, ctx->root_hi, p.root);
  p.width = ctx->ctl.screen_width; p.height = ctx->ctl.screen_height; p.ssaa = ctx->ctl.ssaa;
  p.focal = (float)(ctx->ctl.screen_width * ctx->ctl.ssaa);      // rasteriser.h:17
  p.extinction_c = ctx->ctl.absorption_c + ctx->ctl.scattering_c; // Control.h:48
  p.scattering_c = ctx->ctl.scattering_c;
  // Integral_73 reads the grid when HETEROGENEOUS is set (BeamRadiance.h:159-188)
  const bool generic_het = mode == PB_MODE_GENERIC && ctx->ctl.one_dimensional && ctx->ctl.heterogeneous;
  p.het = ctx->d_het; p.het_axis = ctx->d_het_axis; p.het_fast = ctx->d_het_fast; p.grid_z = (mode == PB_MODE_73HET || generic_het) ? ctx->d_grid : nullptr; p.grid_side = ctx->grid_side; p.grid_fast_div = ctx->grid_fast_div;
  p.one_dimensional = ctx->ctl.one_dimensional;
  p.short_view = ctx->ctl.short_view; p.view_seed = ctx->ctl.seed;
  p.store_d1 = mode != PB_MODE_721 ? 1 : 0;
  if (generic_het && !ctx->d_grid) { pb_set_error("pb_render: PB_MODE_GENERIC with one_dimensional and heterogeneous set needs pb_set_grid"); return PB_ERR_STATE; }
  if (mode == PB_MODE_73HET && !ctx->d_grid) { pb_set_error("pb_render: PB_MODE_73HET needs pb_set_grid"); return PB_ERR_STATE; }
  p.x0 = x0; p.y0 = y0; p.out_w = out_w; p.out_h = out_h; p.strip_stride = strip_stride;
  p.tiles_x = (out_w + 7) / 8; p.tiles_y = (out_h + 7) / 8;
  p.out = d_out;
  if (use_lbvh == 1 && ctx->lb_n >= 2) {
    // the build was queued on the context's stream and not waited for
    if (s != ctx->stream) PB_HIP(hipStreamWaitEvent(s, ctx->ev[5], 0));
    p.lbvh_nodes = ctx->lb_nodes; p.lbvh_sA = ctx->lb_sA; p.lbvh_sB = ctx->lb_sB; p.lbvh_perm = ctx->lb_perm;
    p.lbvh_n = ctx->lb_n;
  }
  p.precise = ctx->precise ? 1 : 0;
  // a rank's share of interleaved strips has few items: shorter ones keep the longest wave short
  p.slice = ctx->tune_slice > 0 ? ctx->tune_slice : (strip_stride > 1 ? 96 : 128);   // render_binned refines it per frame geometry
  p.slice_whole = 0;
  p.xcd_run = -1;           // chosen with the item count in render_binned
  if (pb_tune().xcdrun >= 0 && pb_tune().xcdrun <= 4096) p.xcd_run = pb_tune().xcdrun;
  if (pb_tune().whole >= 0 && pb_tune().whole <= PB_SLICE) p.slice_whole = pb_tune().whole;
  p.eval_waves = ctx->tune_waves;
  p.debug = pb_tune().debug;          // 0 unless the library was built with -DPB_EXPERIMENTS
  p.counters = ctx->count_on ? ctx->d_counters : nullptr;
  if (ctx->count_on) PB_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * PB_CNT_N, s));

  // The chunk pool must hold every candidate chunk of this frame.  Its size depends on the segment set
  // and the frame geometry only, so it is measured (one host synchronisation) the first time a given
  // combination is rendered and trusted afterwards.
  const int groups = p.tiles_x * p.tiles_y * p.ssaa * p.ssaa;
  const unsigned long long key[6] = {ctx->segs_version,
                                     ((unsigned long long)(uint32_t)x0 << 32) | (uint32_t)y0,
                                     ((unsigned long long)(uint32_t)out_w << 32) | (uint32_t)out_h,
                                     (unsigned long long)strip_stride,
                                     (unsigned long long)use_lbvh,
                                     (unsigned long long)mode};
  if (use_lbvh == 2) return render_binned(ctx, ws, p, mode, key, s);
  bool verified = std::memcmp(key, ws.ws_key, sizeof(key)) == 0;
  if (!verified) { ws.queue_kept = false; std::memset(ws.ws_key, 0, sizeof(ws.ws_key)); ws.ws_items_needed = 0; }   // as in render_binned
  int want_chunks = ws.ws_chunks;
  if (!verified && want_chunks < groups * 2 + 1024) want_chunks = groups * 2 + 1024;
  for (int attempt = 0; attempt < 4; ++attempt) {
    int rc = ensure_workspace(ws, groups, want_chunks);
    if (rc != PB_OK) return rc;
    p.rays = ws.d_rays; p.group_last = ws.d_group_last; p.chunks = ws.d_chunks;
    p.partials = ws.d_partials; p.ctrl = ws.d_ctrl; p.chunk_capacity = ws.ws_chunks;
    PB_HIP(hipMemsetAsync(ws.d_ctrl, 0, sizeof(uint32_t) * 4, s));
    if (ctx->count_on || ctx->time_render) PB_HIP(hipEventRecord(ctx->ev[0], s));
    rc = pb_launch_cull(p, mode, ctx->count_on, s);
    if (rc != PB_OK) return rc;
    if (verified) break;
    PB_HIP(hipMemcpyAsync(ctx->h_ctrl, ws.d_ctrl, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
    PB_HIP(hipStreamSynchronize(s));
    const uint32_t needed = ctx->h_ctrl[0];
    if (needed <= (uint32_t)ws.ws_chunks) {
      ws.ws_items_needed = (int)needed;
      ws.queue_kept = false;
      std::memcpy(ws.ws_key, key, sizeof(key));
      verified = true;
      break;
    }
    want_chunks = (int)(needed + needed / 4 + 64);
    if (ctx->count_on) PB_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * PB_CNT_N, s));
  }
  if (!verified) { pb_set_error("pb_render: candidate chunk pool could not be sized"); return PB_ERR_CAPACITY; }
  p.n_items_host = ws.ws_items_needed;
  int rc = pb_launch_eval_reduce(p, mode, ctx->count_on, s);
  if (rc != PB_OK) return rc;
  if (ctx->count_on || ctx->time_render) { PB_HIP(hipEventRecord(ctx->ev[1], s)); ctx->ev_valid[0] = true; }
  return PB_OK;
}

}  // namespace

extern "C" {

void pb_control_defaults(pb_control* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->screen_width = 512; c->screen_height = 512; c->ssaa = 1;
  c->photon_number = 1000; c->bounces = 0; c->adaptive = 0;
  c->short_beams = 0; c->short_view = 0; c->scatter = 1;
  c->one_dimensional = 0; c->fixed_radius = 0; c->heterogeneous = 0;
  c->radius = 0.01f; c->absorbed = 0.0f; c->seed = 42;
  c->light_position[0] = 0; c->light_position[1] = (float)-0.8; c->light_position[2] = (float)-0.5; c->light_position[3] = 1;
  c->light_power[0] = c->light_power[1] = c->light_power[2] = 0.5f * 1;
  c->absorption_c = (float)0.035; c->scattering_c = (float)0.005;
  c->camera[0] = 0; c->camera[1] = 0; c->camera[2] = -3; c->camera[3] = 1.0f;
  c->theta[0] = c->theta[1] = c->theta[2] = 0.0f;
}

const char* pb_last_error(void) { return g_last_error.c_str(); }

int pb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pb_create(const pb_control* control, int device, pb_ctx** out) {
  if (!control || !out) { pb_set_error("pb_create: null argument"); return PB_ERR_ARG; }
  if (control->screen_width <= 0 || control->screen_height <= 0 || control->ssaa <= 0) {
    pb_set_error("pb_create: screen_width, screen_height and ssaa must be positive"); return PB_ERR_ARG;
  }
  int n = pb_device_count();
  if (n <= 0) { pb_set_error("pb_create: no GPU visible; this library has no CPU path"); return PB_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { pb_set_error("pb_create: device index out of range"); return PB_ERR_ARG; }
  pb_ctx* ctx = new pb_ctx();
  ctx->ctl = *control;
  ctx->device = device;
  { static std::mutex mu; static unsigned long long next_id = 0; std::lock_guard<std::mutex> lock(mu); ctx->id = ++next_id; }
  std::memset(&ctx->stats, 0, sizeof(ctx->stats));
  *out = nullptr;
  auto set_up = [&]() -> int {
    PB_HIP(hipSetDevice(device));
    PB_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i) PB_HIP(hipEventCreate(&ctx->ev[i]));
    PB_HIP(hipEventCreateWithFlags(&ctx->ev_readers, hipEventDisableTiming));
    PB_HIP(hipMalloc(&ctx->d_counters, sizeof(unsigned long long) * PB_CNT_N));
    PB_HIP(hipMemset(ctx->d_counters, 0, sizeof(unsigned long long) * PB_CNT_N));
    PB_HIP(hipMalloc(&ctx->d_matrix, sizeof(float) * 16));
    PB_HIP(hipMalloc(&ctx->ws.d_ctrl, sizeof(uint32_t) * 4));
    { const int rc_static = static_workspace_init(device); if (rc_static != PB_OK) return rc_static; }
    PB_HIP(hipHostMalloc(&ctx->h_ctrl, sizeof(uint32_t) * 4));
    return PB_OK;
  };
  const int rc_set_up = set_up();
  if (rc_set_up != PB_OK) { pb_destroy(ctx); return rc_set_up; }      // releases whatever was created
  pb_tune_reload();                   // a context created after the environment changed sees the change
  { const int v = pb_tune().slice; if (v >= 32 && v <= PB_SLICE) ctx->tune_slice = v; }
  { const int v = pb_tune().waves; if (v == 0 || (v >= 64 && v <= (1 << 20))) ctx->tune_waves = v; }
  *out = ctx;
  return PB_OK;
}

void pb_destroy(pb_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  void* ptrs[] = {ctx->d_tri_xyz, ctx->d_matrix, ctx->d_tris, ctx->d_segs, ctx->d_frames, ctx->d_capA, ctx->d_capB,
                  ctx->d_het, ctx->d_het_axis, ctx->d_het_fast, ctx->d_grid, ctx->d_out, ctx->d_counters, ctx->d_beams, ctx->d_beams_spare, ctx->d_emit_tmp[0],
                  ctx->d_emit_tmp[1], ctx->d_emit_tmp[2], ctx->ws.d_rays, ctx->ws.d_group_last,
                  ctx->ws.d_chunks, ctx->ws.d_partials, ctx->ws.d_ctrl, ctx->ws.d_tile_u32, ctx->ws.d_bin_list, ctx->ws.d_blk,
                  ctx->ws.d_items, ctx->ws.d_accum, ctx->ws.d_scan_tmp, ctx->ws.d_walk, ctx->ws.d_residue, ctx->ws.d_surv_list,
                  ctx->ws.d_surv_cnt, ctx->ws.d_items_kept, ctx->lb_pool};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (ctx->h_ctrl) (void)hipHostFree(ctx->h_ctrl);
  for (int i = 0; i < 8; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  if (ctx->ev_readers) (void)hipEventDestroy(ctx->ev_readers);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void pb_transformation_matrix(const float theta[3], const float camera[4], float M[16]) {
  // skeleton.cpp:512-532, column-major M[col*4+row]
  float cx = std::cos(theta[0]), sx = std::sin(theta[0]);
  float cy = std::cos(theta[1]), sy = std::sin(theta[1]);
  float cz = std::cos(theta[2]), sz = std::sin(theta[2]);
  M[0] = cy * cz; M[1] = cy * sz; M[2] = -sy; M[3] = 0;
  M[4] = (-cx * sz) + (sx * sy * cz); M[5] = (cx * cz) + (sx * sy * sz); M[6] = sx * cy; M[7] = 0;
  M[8] = (sx * sz) + (cx * sy * cz); M[9] = (-sx * cz) + (cx * sy * sz); M[10] = cx * cy; M[11] = 0;
  M[12] = -camera[0]; M[13] = -camera[1]; M[14] = -camera[2]; M[15] = 1;
}

int pb_set_scene(pb_ctx* ctx, const float* tri_xyz, int n_tri, const float matrix[16]) {
  if (!ctx || !tri_xyz || !matrix || n_tri <= 0) { pb_set_error("pb_set_scene: bad argument"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  if (ctx->d_tri_xyz) { PB_HIP(hipFree(ctx->d_tri_xyz)); ctx->d_tri_xyz = nullptr; }
  if (ctx->d_tris) { PB_HIP(hipFree(ctx->d_tris)); ctx->d_tris = nullptr; }
  PB_HIP(hipMalloc(&ctx->d_tri_xyz, sizeof(float) * 9 * (size_t)n_tri));
  PB_HIP(hipMalloc(&ctx->d_tris, (sizeof(TriPrep) + sizeof(float4) + 12 * sizeof(float)) * (size_t)n_tri));
  PB_HIP(hipMemcpyAsync(ctx->d_tri_xyz, tri_xyz, sizeof(float) * 9 * (size_t)n_tri, hipMemcpyHostToDevice, ctx->stream));
  PB_HIP(hipMemcpyAsync(ctx->d_matrix, matrix, sizeof(float) * 16, hipMemcpyHostToDevice, ctx->stream));
  rc = pb_launch_prepare_scene(ctx->d_tri_xyz, n_tri, ctx->d_matrix, ctx->d_tris, (float4*)(ctx->d_tris + n_tri),
                               (float*)((float4*)(ctx->d_tris + n_tri) + n_tri), ctx->stream);
  if (rc != PB_OK) return rc;
  PB_HIP(hipStreamSynchronize(ctx->stream));
  ctx->n_tri = n_tri;
  ctx->segs_version++;
  std::memcpy(ctx->matrix, matrix, sizeof(float) * 16);
  return PB_OK;
}

// DdaWalk::sample divides by the cell size c = 2 / (side - 1) twice per sample.  With r = RN(1 / c) the
// sequence q = RN(x r), q' = RN(q + RN(x - q c) r) (both RN(.) of the second step are fused multiply-adds, the
// inner one exact) gives the correctly rounded x / c for all x of a divisor, or fails for a few x, depending on
// c; scaling x by a power of two scales every step exactly, so trying the 2^23 significands of one binade
// settles it for every x whose intermediate results stay normal.  True: the kernels may use the sequence.
static bool known_divisor_rounds_correctly(int side) {
  static std::mutex mu;
  static std::map<int, bool> checked;
  std::lock_guard<std::mutex> lock(mu);
  auto it = checked.find(side);
  if (it != checked.end()) return it->second;
  const float res = (float)(side - 1);
  const volatile float c = 2.0f / res;
  const float cc = c, r = 1.0f / cc;
  bool ok = std::isfinite(r) && cc >= 0x1p-20f;
  for (uint32_t m = 1u << 23; ok && m < (1u << 24); ++m) {
    const float x = (float)m;
    const float q = x * r;
    const float q1 = std::fmaf(std::fmaf(-q, cc, x), r, q);
    ok = q1 == x / cc;
  }
  checked[side] = ok;
  return ok;
}

int pb_set_grid(pb_ctx* ctx, const float* sigma_t, int side) {
  if (!ctx || !sigma_t || side < 2 || side > 4096) { pb_set_error("pb_set_grid: bad argument (2 <= side <= 4096)"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  const size_t n = (size_t)side * side;
  // the field, then (16-byte aligned) the four values of every cell index: GridView::cells() in pb_grid.h
  const size_t n_pad = (n + 3) & ~(size_t)3;
  std::vector<float> host(n_pad + 4 * n, 0.0f);
  std::memcpy(host.data(), sigma_t, sizeof(float) * n);
  for (int row = 0; row < side; ++row)
    for (int col = 0; col < side; ++col) {
      const size_t i = (size_t)row * side + col;
      float* c = host.data() + n_pad + 4 * i;
      c[0] = sigma_t[i];
      c[1] = (row + 1 < side) ? sigma_t[i + side] : sigma_t[i];
      c[2] = (col + 1 < side) ? sigma_t[i + 1] : sigma_t[i];
      c[3] = c[1];
    }
  if (host.size() > ctx->grid_cap_floats) {
    ctx->grid_side = 0; ctx->grid_fast_div = 0; ctx->grid_cap_floats = 0;
    if (ctx->d_grid) { PB_HIP(hipFree(ctx->d_grid)); ctx->d_grid = nullptr; }
    PB_HIP(hipMalloc(&ctx->d_grid, sizeof(float) * host.size()));
    ctx->grid_cap_floats = host.size();
  } else {
    // the buffer is overwritten in place: frames still in flight on any stream read the field they were launched with
    PB_HIP(hipDeviceSynchronize());
  }
  PB_HIP(hipMemcpyAsync(ctx->d_grid, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice, ctx->stream));
  PB_HIP(hipStreamSynchronize(ctx->stream));
  ctx->grid_side = side;
  ctx->grid_fast_div = known_divisor_rounds_correctly(side) ? 1 : 0;
  return PB_OK;
}

static int upload_common(pb_ctx* ctx, const pb_segment* src, bool src_on_device, int n, const float root_lo[3],
                         const float root_hi[3], hipStream_t s) {
  if (!ctx || (!src && n > 0) || n < 0 || !root_lo || !root_hi) { pb_set_error("pb_upload_segments: bad argument"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  rc = pb_ensure_segment_capacity(ctx, n);
  if (rc != PB_OK) return rc;
  rc = pb_wait_for_readers(ctx, s, false);      // frames still reading the old records, on whatever stream
  if (rc != PB_OK) return rc;
  if (n > 0)
    PB_HIP(hipMemcpyAsync(ctx->d_segs, src, sizeof(pb_segment) * (size_t)n,
                          src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  ctx->n_seg = n;
  for (int k = 0; k < 3; ++k) { ctx->root_lo[k] = root_lo[k]; ctx->root_hi[k] = root_hi[k]; }
  rc = pb_prepare_frames(ctx, default_mode(ctx), s);
  if (rc != PB_OK) return rc;
  ctx->have_segments = true;
  ctx->segs_version++;
  return PB_OK;
}

int pb_upload_segments(pb_ctx* ctx, const pb_segment* segs, int n, const float root_lo[3], const float root_hi[3]) {
  if (!ctx) { pb_set_error("pb_upload_segments: null context"); return PB_ERR_ARG; }
  int rc = upload_common(ctx, segs, false, n, root_lo, root_hi, ctx->stream);
  if (rc != PB_OK) return rc;
  PB_HIP(hipStreamSynchronize(ctx->stream));
  return PB_OK;
}

int pb_upload_segments_d(pb_ctx* ctx, const pb_segment* d_segs, int n, const float root_lo[3], const float root_hi[3],
                         void* stream) {
  return upload_common(ctx, d_segs, true, n, root_lo, root_hi, (hipStream_t)stream);
}

int pb_segment_count(pb_ctx* ctx, int* n_out) {
  if (!ctx || !n_out) { pb_set_error("pb_segment_count: null argument"); return PB_ERR_ARG; }
  *n_out = ctx->have_segments ? ctx->n_seg : 0;
  return PB_OK;
}

int pb_download_segments(pb_ctx* ctx, pb_segment* out, int capacity, int* n_out) {
  if (!ctx || !out || !n_out) { pb_set_error("pb_download_segments: null argument"); return PB_ERR_ARG; }
  if (!ctx->have_segments) { pb_set_error("pb_download_segments: no segment set"); return PB_ERR_STATE; }
  if (capacity < ctx->n_seg) { pb_set_error("pb_download_segments: output array too small"); return PB_ERR_CAPACITY; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  PB_HIP(hipDeviceSynchronize());
  PB_HIP(hipMemcpy(out, ctx->d_segs, sizeof(pb_segment) * (size_t)ctx->n_seg, hipMemcpyDeviceToHost));
  *n_out = ctx->n_seg;
  return PB_OK;
}

int pb_save_segments(pb_ctx* ctx, const char* path) {
  if (!ctx || !path) { pb_set_error("pb_save_segments: null argument"); return PB_ERR_ARG; }
  if (!ctx->have_segments) { pb_set_error("pb_save_segments: no segment set"); return PB_ERR_STATE; }
  std::vector<pb_segment> segs((size_t)ctx->n_seg);
  int n = 0;
  int rc = pb_download_segments(ctx, segs.data(), ctx->n_seg, &n);
  if (rc != PB_OK) return rc;
  return pb_write_segments_file(path, segs.data(), n, ctx->root_lo, ctx->root_hi);
}

int pb_load_segments(pb_ctx* ctx, const char* path) {
  if (!ctx || !path) { pb_set_error("pb_load_segments: null argument"); return PB_ERR_ARG; }
  int n = 0;
  float lo[3], hi[3];
  int rc = pb_read_segments_file(path, nullptr, 0, &n, lo, hi);
  if (rc != PB_OK) return rc;
  std::vector<pb_segment> segs((size_t)n);
  rc = pb_read_segments_file(path, segs.data(), n, &n, lo, hi);
  if (rc != PB_OK) return rc;
  return pb_upload_segments(ctx, segs.data(), n, lo, hi);
}

int pb_get_root_box(pb_ctx* ctx, float root_lo[3], float root_hi[3]) {
  if (!ctx || !root_lo || !root_hi) { pb_set_error("pb_get_root_box: null argument"); return PB_ERR_ARG; }
  if (!ctx->have_segments) { pb_set_error("pb_get_root_box: no segment set"); return PB_ERR_STATE; }
  for (int k = 0; k < 3; ++k) { root_lo[k] = ctx->root_lo[k]; root_hi[k] = ctx->root_hi[k]; }
  return PB_OK;
}

int pb_emit_beams(pb_ctx* ctx, int n_photons, uint64_t seed, int* n_beams_out, int* n_segments_out) {
  if (!ctx || n_photons < 0) { pb_set_error("pb_emit_beams: bad argument"); return PB_ERR_ARG; }
  if (!ctx->d_tris) { pb_set_error("pb_emit_beams: pb_set_scene has not been called"); return PB_ERR_STATE; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  rc = pb_emit_run(ctx, n_photons, seed, false);
  if (rc != PB_OK) return rc;
  if (n_beams_out) *n_beams_out = ctx->n_beams_const + ctx->n_beams_scatt;
  if (n_segments_out) *n_segments_out = ctx->n_seg;
  return PB_OK;
}

int pb_recast_beams(pb_ctx* ctx, uint64_t seed, int* n_scattered_out, int* n_segments_out) {
  if (!ctx) { pb_set_error("pb_recast_beams: null context"); return PB_ERR_ARG; }
  if (!ctx->d_tris) { pb_set_error("pb_recast_beams: pb_set_scene has not been called"); return PB_ERR_STATE; }
  if (!ctx->d_beams) { pb_set_error("pb_recast_beams: pb_emit_beams has not been called"); return PB_ERR_STATE; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  rc = pb_emit_run(ctx, ctx->n_beams_const, seed, true);
  if (rc != PB_OK) return rc;
  if (n_scattered_out) *n_scattered_out = ctx->n_beams_scatt;
  if (n_segments_out) *n_segments_out = ctx->n_seg;
  return PB_OK;
}

int pb_download_beams(pb_ctx* ctx, pb_beam* out, int capacity, int* n_out) {
  if (!ctx || !out || !n_out) { pb_set_error("pb_download_beams: null argument"); return PB_ERR_ARG; }
  const int n = ctx->n_beams_const + ctx->n_beams_scatt;
  if (!ctx->d_beams && n > 0) { pb_set_error("pb_download_beams: pb_emit_beams has not been called"); return PB_ERR_STATE; }
  if (capacity < n) { pb_set_error("pb_download_beams: output array too small"); return PB_ERR_CAPACITY; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  PB_HIP(hipDeviceSynchronize());
  if (n > 0) PB_HIP(hipMemcpy(out, ctx->d_beams, sizeof(pb_beam) * (size_t)n, hipMemcpyDeviceToHost));
  *n_out = n;
  return PB_OK;
}

int pb_beam_counts(pb_ctx* ctx, int* n_const, int* n_scatt) {
  if (!ctx || !n_const || !n_scatt) { pb_set_error("pb_beam_counts: null argument"); return PB_ERR_ARG; }
  *n_const = ctx->n_beams_const; *n_scatt = ctx->n_beams_scatt;
  return PB_OK;
}

int pb_build_lbvh(pb_ctx* ctx) {
  if (!ctx) { pb_set_error("pb_build_lbvh: null context"); return PB_ERR_ARG; }
  if (!ctx->have_segments) { pb_set_error("pb_build_lbvh: no segment set"); return PB_ERR_STATE; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  return pb_lbvh_build(ctx);
}

int pb_download_lbvh(pb_ctx* ctx, uint64_t* keys, uint32_t* perm, pb_node* nodes) {
  if (!ctx) { pb_set_error("pb_download_lbvh: null context"); return PB_ERR_ARG; }
  if (ctx->lb_n < 0 || ctx->lb_version != ctx->segs_version) { pb_set_error("pb_download_lbvh: pb_build_lbvh has not been called for this segment set"); return PB_ERR_STATE; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  return pb_lbvh_download(ctx, keys, perm, nodes);
}

int pb_render_d(pb_ctx* ctx, int x0, int y0, int w, int h, int mode, int use_lbvh, float* d_rgb_out, void* stream) {
  if (!ctx || !d_rgb_out || w <= 0 || h <= 0 || x0 < 0 || y0 < 0) { pb_set_error("pb_render: bad argument"); return PB_ERR_ARG; }
  if (x0 + w > ctx->ctl.screen_width || y0 + h > ctx->ctl.screen_height) { pb_set_error("pb_render: rectangle exceeds the screen"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  return render_common(ctx, ctx->ws, x0, y0, w, h, 1, mode, use_lbvh, d_rgb_out, (hipStream_t)stream);
}

int pb_render_noalloc_d(pb_ctx* ctx, int x0, int y0, int w, int h, int mode, float* d_rgb_out, void* stream) {
  if (!ctx || !d_rgb_out || w <= 0 || h <= 0 || x0 < 0 || y0 < 0) { pb_set_error("pb_render_noalloc_d: bad argument"); return PB_ERR_ARG; }
  if (x0 + w > ctx->ctl.screen_width || y0 + h > ctx->ctl.screen_height) { pb_set_error("pb_render_noalloc_d: rectangle exceeds the screen"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  if (ctx->device < 0 || ctx->device >= MAX_DEVICES || !g_static[ctx->device].base) {
    pb_set_error("pb_render_noalloc_d: the static workspace of this GPU was not set up by pb_create"); return PB_ERR_STATE;
  }
  // sizes are checked against the stated maxima before anything is launched
  const long long groups = (long long)((w + 7) / 8) * ((h + 7) / 8) * ctx->ctl.ssaa * ctx->ctl.ssaa;
  if (groups > PB_NOALLOC_MAX_GROUPS) {
    pb_set_error("pb_render_noalloc_d: " + std::to_string(groups) + " ray groups (8x8-pixel tiles x SSAA^2) exceed PB_NOALLOC_MAX_GROUPS = " +
                 std::to_string(PB_NOALLOC_MAX_GROUPS));
    return PB_ERR_CAPACITY;
  }
  return render_common(ctx, g_static[ctx->device].ws, x0, y0, w, h, 1, mode, 2, d_rgb_out, (hipStream_t)stream);
}

int pb_render(pb_ctx* ctx, int x0, int y0, int w, int h, int mode, int use_lbvh, float* rgb_out) {
  if (!ctx || !rgb_out || w <= 0 || h <= 0) { pb_set_error("pb_render: bad argument"); return PB_ERR_ARG; }
  if (x0 < 0 || y0 < 0 || x0 + w > ctx->ctl.screen_width || y0 + h > ctx->ctl.screen_height) { pb_set_error("pb_render: rectangle exceeds the screen"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  size_t need = (size_t)w * h * 3;
  if (need > ctx->out_floats) {
    if (ctx->d_out) { PB_HIP(hipFree(ctx->d_out)); ctx->d_out = nullptr; ctx->out_floats = 0; }
    PB_HIP(hipMalloc(&ctx->d_out, sizeof(float) * need));
    ctx->out_floats = need;
  }
  ctx->time_render = true;       // this call waits for the frame anyway: its device time is recorded (pb_stats.ms_render)
  rc = pb_render_d(ctx, x0, y0, w, h, mode, use_lbvh, ctx->d_out, ctx->stream);
  ctx->time_render = false;
  if (rc != PB_OK) return rc;
  PB_HIP(hipMemcpyAsync(rgb_out, ctx->d_out, sizeof(float) * need, hipMemcpyDeviceToHost, ctx->stream));
  PB_HIP(hipStreamSynchronize(ctx->stream));
  return PB_OK;
}

int pb_render_strips_d(pb_ctx* ctx, int first_strip, int strip_stride, int n_strips, int mode, int use_lbvh,
                       float* d_rgb_out, void* stream) {
  if (!ctx || !d_rgb_out || first_strip < 0 || strip_stride <= 0 || n_strips <= 0) { pb_set_error("pb_render_strips_d: bad argument"); return PB_ERR_ARG; }
  const int last_y = 8 * (first_strip + (n_strips - 1) * strip_stride);
  if (last_y >= ctx->ctl.screen_height) { pb_set_error("pb_render_strips_d: strips exceed the screen"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  return render_common(ctx, ctx->ws, 0, 8 * first_strip, ctx->ctl.screen_width, 8 * n_strips, strip_stride, mode, use_lbvh,
                       d_rgb_out, (hipStream_t)stream);
}

int pb_quantise_argb(const float* rgb, int n_pixels, uint32_t* argb_out) {
  if (!rgb || !argb_out || n_pixels < 0) { pb_set_error("pb_quantise_argb: bad argument"); return PB_ERR_ARG; }
  for (int i = 0; i < n_pixels; ++i) {
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
      float v = 255 * rgb[3 * (size_t)i + k];
      v = (v < 0.f) ? 0.f : v;          // glm::clamp = min(max(x, lo), hi)
      v = (255.f < v) ? 255.f : v;
      q[k] = (uint32_t)v;
    }
    argb_out[i] = (128u << 24) + (q[0] << 16) + (q[1] << 8) + q[2];
  }
  return PB_OK;
}

void pb_reload_tuning(void) { pb_tune_reload(); }

int pb_set_precise(pb_ctx* ctx, int on) {
  if (!ctx) { pb_set_error("pb_set_precise: null context"); return PB_ERR_ARG; }
  ctx->precise = on != 0;
  return PB_OK;
}

int pb_enable_counters(pb_ctx* ctx, int on) {
  if (!ctx) { pb_set_error("pb_enable_counters: null context"); return PB_ERR_ARG; }
  ctx->count_on = on != 0;
  return PB_OK;
}

int pb_counters(pb_ctx* ctx, pb_stats* out) {
  if (!ctx || !out) { pb_set_error("pb_counters: null argument"); return PB_ERR_ARG; }
  int rc = ensure_device(ctx);
  if (rc != PB_OK) return rc;
  PB_HIP(hipDeviceSynchronize());
  float ms = 0;
  if (ctx->ev_valid[0]) { PB_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); ctx->stats.ms_render = ms; }
  if (ctx->ev_valid[1]) { PB_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->stats.ms_prepare = ms; }
  if (ctx->ev_valid[2]) { PB_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->stats.ms_build = ms; }
  if (ctx->ev_valid[3]) { PB_HIP(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7])); ctx->stats.ms_emit = ms; }
  unsigned long long h[PB_CNT_N];
  PB_HIP(hipMemcpy(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost));
  ctx->stats.rays = h[PB_CNT_RAYS]; ctx->stats.rays_hit_scene = h[PB_CNT_RAYS_HIT];
  ctx->stats.candidates = h[PB_CNT_CAND]; ctx->stats.cyl_hits = h[PB_CNT_CYL];
  ctx->stats.valid_hits = h[PB_CNT_VALID]; ctx->stats.nodes_visited = h[PB_CNT_NODES];
  ctx->stats.gate_passes = h[PB_CNT_GATED];
  ctx->stats.shortcut_wrong = h[PB_CNT_SHORTCUT_WRONG];
  if (pb_tune().debug_times && h[PB_CNT_T_END]) {
    const double t0 = (double)~h[PB_CNT_T_START_INV], td = (double)~h[PB_CNT_T_DRAIN_INV], te = (double)h[PB_CNT_T_END];
    std::fprintf(stderr, "[pb] evaluation queue: %llu items, first wave leaves the empty queue at %.1f us, last wave ends at %.1f us\n",
                 (unsigned long long)h[PB_CNT_ITEMS], (td - t0) * 0.01, (te - t0) * 0.01);
  }
  *out = ctx->stats;
  return PB_OK;
}

}  // extern "C"
