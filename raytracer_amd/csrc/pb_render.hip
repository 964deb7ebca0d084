// DrawStandard (skeleton.cpp:330-365) on the GPU: one wavefront owns an 8x8
// pixel tile (one camera ray per lane).  All camera rays start at (0,0,0), so
// the rays of a tile span a thin pyramid with its apex at the origin; the wave
// first finds, cooperatively (one segment or one LBVH node per lane), the
// segments whose capsule can touch that pyramid, then every lane evaluates its
// own ray against each candidate, whose per-segment record is wave-uniform
// and therefore fetched through the scalar unit.
//
// Built with -ffp-contract=off: the gate (HitBoundingBox + max_distance,
// BeamRadiance.h:426-430) is discontinuous in its inputs and is evaluated with
// the reference's exact operation order.
#include <algorithm>
#include <cstdlib>

#include "pb_internal.h"
#include "pb_geom.h"
#include "pb_grid.h"
#include "pb_pair.h"

namespace {

// pl[k] are the four inward unit normals of the side planes (through the origin) of a tile's pyramid.
// Is some point of the axis a-b within `slack` of all four half-spaces at once?  Clips the parameter
// interval [0,1] by each plane; tighter than testing the end points plane by plane.
PB_DI bool axis_touches_pyramid(const float4 A, const float4 B, const V3 pl[4]) {
  const float slack = A.w * 1.001f + 1e-5f;
  float t0 = -1e-5f, t1 = 1.0f + 1e-5f;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float fa = pl[k].x * A.x + pl[k].y * A.y + pl[k].z * A.z + slack;
    const float fb = pl[k].x * B.x + pl[k].y * B.y + pl[k].z * B.z + slack;
    ok = ok && !(fa < 0.0f && fb < 0.0f);
    const float t = fa * fast_rcp(fa - fb);          // only used when fa and fb differ in sign
    if (fa < 0.0f && fb >= 0.0f) t0 = fmaxf(t0, t - 1e-5f);
    if (fb < 0.0f && fa >= 0.0f) t1 = fminf(t1, t + 1e-5f);
  }
  return ok && (t0 <= t1);
}

// ---------------------------------------------------------------------------
// Ordered frame (cull modes 0 and 1) = three launches.
//  cull           per ray group (an 8x8 tile, one sub-sample): sets up the 64 rays, stores them, finds the
//                 segments whose axis can come within their radius of the tile pyramid - cull_scan_kernel
//                 tests every segment, one per lane; cull_lbvh_block_kernel walks the LBVH, one node per
//                 lane - and appends their indices to chunks of PB_CHUNK candidates taken from a pool.
//  eval_kernel    one wave per chunk: evaluates the 64 rays of the chunk's group against the chunk's
//                 candidates and stores 64 x 3 partial sums.  A tile over the light source has two orders
//                 of magnitude more candidates than the median tile; cutting it into chunks is what keeps
//                 one wave from serialising it.
//  reduce_kernel  one wave per tile adds the partial sums of the tile's chunks in list order (the same
//                 order every run) and writes current / SSAA.
// ---------------------------------------------------------------------------

PB_DI uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// An item's candidates against its tile's pyramid, 64 at a time; the survivors' indices are compacted into `surv` (LDS) and
// counted.  AHEAD, for the kernels that run four waves to a SIMD and wait for memory here, one trip after the other: the
// next 64 list entries are requested before the current ones are tested, and a candidate's two cap records are requested
// together (left to itself the compiler waits for the first before it asks for the second).  At eight waves to a SIMD the
// plain loop is faster (512^2, 17 424 segments: 0.1937 against 0.1952 ms; 0.1955 with only the caps requested together).
// refined (wave-uniform, the item's fourth word): the item's list already holds survivors, copied as they are.
template <bool AHEAD>
PB_DI int refine_candidates(const RenderParams& p, uint32_t start, int cn, const V3 pl[4], uint32_t* surv, int lane, uint32_t refined) {
  int ns = 0;
  if (refined) {
    const uint32_t* kept = p.surv_list + start;
    for (int base = 0; base < cn; base += 64) { const int k = base + lane; if (k < cn) surv[k] = kept[k]; }
    return cn;
  }
  const uint32_t* list = p.bin_list + start;
  if (!AHEAD) {
    for (int base = 0; base < cn; base += 64) {
      const int k = base + lane;
      const uint32_t cand = (k < cn) ? list[k] : 0xffffffffu;
      bool touch = false;
      if (cand != 0xffffffffu) touch = axis_touches_pyramid(p.capA[cand], p.capB[cand], pl);
      const unsigned long long mask = __ballot(touch);
      if (touch) surv[ns + (int)lanes_below(mask)] = cand;
      ns += __builtin_popcountll(mask);
    }
    return ns;
  }
  uint32_t cand = (lane < cn) ? list[lane] : 0xffffffffu;
  for (int base = 0; base < cn; base += 64) {
    const uint32_t at = (cand != 0xffffffffu) ? cand : 0u;        // lanes past the list read segment 0 and drop it
    float4 A = p.capA[at], B = p.capB[at];
    const int kn = base + 64 + lane;
    // stays in flight while the caps are waited for (read by every lane, so that the wait can tell the loads apart: past
    // the list's end the last entry, dropped below)
    const uint32_t next_read = list[kn < cn ? kn : cn - 1];
    asm volatile("" : "+v"(A.x), "+v"(A.y), "+v"(A.z), "+v"(A.w), "+v"(B.x), "+v"(B.y), "+v"(B.z));
    const bool touch = (cand != 0xffffffffu) && axis_touches_pyramid(A, B, pl);
    const unsigned long long mask = __ballot(touch);
    if (touch) surv[ns + (int)lanes_below(mask)] = cand;
    ns += __builtin_popcountll(mask);
    cand = (kn < cn) ? next_read : 0xffffffffu;
  }
  return ns;
}

struct ChunkWriter {
  uint32_t* buf;      // LDS, PB_CHUNK + 64 entries
  uint32_t count;     // wave-uniform
  int prev;           // last chunk written for this group, -1 if none
  int group;
  uint32_t next = 0, left = 0, batch = 1;   // chunk ids reserved from the pool and not used yet
};

// Chunk ids come from one device-wide counter, and same-address atomics run at about 12 ns each on this
// chip: a group reserves 1, 2, 4, ... ids at a time (one atomic per reservation, at most as many ids unused
// as used).  Unused ids are closed as empty chunks, which eval_kernel skips.
PB_DI void close_unused_chunks(const RenderParams& p, int group, uint32_t next, uint32_t left, int tid, int n_threads) {
  for (uint32_t k = tid; k < left; k += n_threads) {
    const uint32_t c = next + k;
    if (c < (uint32_t)p.chunk_capacity) {
      uint32_t* dst = p.chunks + (size_t)c * PB_CHUNK_WORDS;
      dst[0] = (uint32_t)group; dst[1] = 0u; dst[2] = 0xffffffffu; dst[3] = 0u;
    }
  }
}

PB_DI void flush_chunk(const RenderParams& p, ChunkWriter& w, uint32_t n, int lane) {
  if (w.left == 0) {
    uint32_t first = 0;
    if (lane == 0) first = atomicAdd(&p.ctrl[0], w.batch);
    w.next = __builtin_amdgcn_readfirstlane(first);
    w.left = w.batch;
    w.batch *= 2;
  }
  const uint32_t c = w.next;
  w.next++; w.left--;
  if (c < (uint32_t)p.chunk_capacity) {
    uint32_t* dst = p.chunks + (size_t)c * PB_CHUNK_WORDS;
    if (lane == 0) { dst[0] = (uint32_t)w.group; dst[1] = n; dst[2] = (uint32_t)w.prev; dst[3] = 0u; }
    for (uint32_t k = lane; k < n; k += 64) dst[4 + k] = w.buf[k];
    w.prev = (int)c;
  }
}

// append the set lanes' values; whenever PB_CHUNK candidates are buffered, emit a chunk
PB_DI void append_candidates(const RenderParams& p, ChunkWriter& w, bool take, uint32_t value, int lane) {
  const unsigned long long mask = __ballot(take);
  if (mask == 0ull) return;
  if (take) w.buf[w.count + lanes_below(mask)] = value;
  w.count += (uint32_t)__builtin_popcountll(mask);
  __syncthreads();
  if (w.count >= PB_CHUNK) {
    flush_chunk(p, w, PB_CHUNK, lane);
    const uint32_t rest = w.count - PB_CHUNK;
    uint32_t keep = 0;
    if ((uint32_t)lane < rest) keep = w.buf[PB_CHUNK + lane];
    __syncthreads();
    if ((uint32_t)lane < rest) w.buf[lane] = keep;
    w.count = rest;
    __syncthreads();
  }
}

// Conservative test of an (already padded) box against the four side planes of the tile pyramid.
PB_DI bool box_touches_pyramid(const float lo[3], const float hi[3], const V3 pl[4]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d = pl[k].x * (pl[k].x >= 0 ? hi[0] : lo[0]) + pl[k].y * (pl[k].y >= 0 ? hi[1] : lo[1]) +
                    pl[k].z * (pl[k].z >= 0 ? hi[2] : lo[2]);
    ok = ok && (d >= 0.0f);
  }
  return ok;
}

// Inward unit normals of the four side planes (through the origin) of the pyramid spanned by all
// sub-rays of tile (tx, tk), widened by half a sub-pixel.
PB_DI void tile_planes(const RenderParams& p, int tx, int tk, V3 pl[4]) {
  const float half_w = (p.width * p.ssaa) / (float)2, half_h = (p.height * p.ssaa) / (float)2;
  const float xl = (float)((p.x0 + 8 * tx) * p.ssaa) - half_w - 0.5f;
  const float xr = (float)((p.x0 + 8 * tx + 7) * p.ssaa + p.ssaa - 1) - half_w + 0.5f;
  const int ty0 = p.y0 + 8 * tk * p.strip_stride;
  const float yt = (float)(ty0 * p.ssaa) - half_h - 0.5f;
  const float yb = (float)((ty0 + 7) * p.ssaa + p.ssaa - 1) - half_h + 0.5f;
  pl[0] = normalize3(v3(p.focal, 0.0f, -xl));    // x/z >= xl/focal
  pl[1] = normalize3(v3(-p.focal, 0.0f, xr));    // x/z <= xr/focal
  pl[2] = normalize3(v3(0.0f, p.focal, -yt));
  pl[3] = normalize3(v3(0.0f, -p.focal, yb));
}
// Plane k of tile_planes alone, bit for bit (k may differ from lane to lane).
PB_DI V3 tile_plane(const RenderParams& p, int tx, int tk, int k) {
  const float half_w = (p.width * p.ssaa) / (float)2, half_h = (p.height * p.ssaa) / (float)2;
  const float xl = (float)((p.x0 + 8 * tx) * p.ssaa) - half_w - 0.5f;
  const float xr = (float)((p.x0 + 8 * tx + 7) * p.ssaa + p.ssaa - 1) - half_w + 0.5f;
  const int ty0 = p.y0 + 8 * tk * p.strip_stride;
  const float yt = (float)(ty0 * p.ssaa) - half_h - 0.5f;
  const float yb = (float)((ty0 + 7) * p.ssaa + p.ssaa - 1) - half_h + 0.5f;
  const float f = (k & 1) ? -p.focal : p.focal;
  const float off = k == 0 ? -xl : (k == 1 ? xr : (k == 2 ? -yt : yb));
  return normalize3(k < 2 ? v3(f, 0.0f, off) : v3(0.0f, f, off));
}

template <bool COUNT>
__global__ __launch_bounds__(64) void cull_scan_kernel(const RenderParams p) {
  __shared__ uint32_t buf[PB_CHUNK + 64];
  const int lane = threadIdx.x;
  const int group = blockIdx.x;
  const int nsub = p.ssaa * p.ssaa;
  const int tile = group / nsub, sub = group % nsub;
  const int si = sub / p.ssaa, sj = sub % p.ssaa;
  const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
  const int lx = lane & 7, ly = lane >> 3;
  const int ox = 8 * tx + lx, oy = 8 * tk + ly;             // output coordinates
  const int px = p.x0 + ox, py = p.y0 + 8 * tk * p.strip_stride + ly;
  const bool in_image = (ox < p.out_w) && (oy < p.out_h) && (px < p.width) && (py < p.height);

  Ray r = make_ray(p, px * p.ssaa + si, py * p.ssaa + sj);
  r.active = r.active && in_image;
  // the root gate is a property of the ray alone (BeamRadiance.h:426-430 at the tree root)
  const bool hit_scene = r.active;
  r.active = r.active && box_gate(p.root.lo, p.root.hi, p.root.cand, p.root.qflags, r.dn, r.maxd);
  if (!r.active) r.maxd = -1.0f;     // every gate fails for this lane
  float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  rr[0 * 64 + lane] = r.l.x; rr[1 * 64 + lane] = r.l.y; rr[2 * 64 + lane] = r.l.z;
  rr[3 * 64 + lane] = r.dn.x; rr[4 * 64 + lane] = r.dn.y; rr[5 * 64 + lane] = r.dn.z;
  rr[6 * 64 + lane] = r.maxd; rr[7 * 64 + lane] = 0.0f;
  rr[8 * 64 + lane] = r.d1.x; rr[9 * 64 + lane] = r.d1.y; rr[10 * 64 + lane] = r.d1.z; rr[11 * 64 + lane] = 0.0f;

  ChunkWriter w;
  w.buf = buf; w.count = 0; w.prev = -1; w.group = group;
  unsigned long long c_gated = 0;
  if (__ballot(r.active) != 0ull) {
    V3 pl[4];
    tile_planes(p, tx, tk, pl);   // wave-uniform
    if (COUNT) {
      // the pairs on which the reference calls HitCylinder: both gates pass, cylinder or not
      for (int sidx = 0; sidx < p.n_seg; ++sidx) {
        const SegFrame& f = p.frames[sidx];
        c_gated += box_gate(f.lo, f.hi, f.cand, f.qflags, r.dn, r.maxd) ? 1 : 0;
      }
    }
    for (int base = 0; base < p.n_seg; base += 64) {
      const int s = base + lane;
      bool touch = false;
      if (s < p.n_seg) touch = axis_touches_pyramid(p.capA[s], p.capB[s], pl);
      append_candidates(p, w, touch, (uint32_t)s, lane);
    }
    if (w.count > 0) flush_chunk(p, w, w.count, lane);
    close_unused_chunks(p, group, w.next, w.left, lane, 64);
  }
  if (lane == 0) p.group_last[group] = w.prev;
  if (COUNT) {
    unsigned long long v[3] = {in_image ? 1ull : 0ull, hit_scene ? 1ull : 0ull, c_gated};
    const int slot[3] = {PB_CNT_RAYS, PB_CNT_RAYS_HIT, PB_CNT_GATED};
    for (int k = 0; k < 3; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&p.counters[slot[k]], x);
    }
  }
}

// LBVH traversal with CULL_WAVES waves per tile.  The length of a one-wave traversal is set by its busiest
// tile: the tile over the light source meets every first-cast beam, and one wave pops those leaves 64 at a
// time.  Here the block's waves share the stack and the candidate buffer: up to CULL_THREADS nodes are popped
// per step, pushes and candidates are compacted across the block (wave ballots, then per-wave offsets
// through LDS), in thread order, so the candidate list is the same on every run.
constexpr int CULL_WAVES = 4;
constexpr int CULL_THREADS = 64 * CULL_WAVES;
constexpr int CULL_BUF = PB_CHUNK + 2 * CULL_THREADS;

struct BlockCounts { int n[2][CULL_WAVES]; };

// exclusive offset of this thread's wave and the block total, for two flag sets at once
PB_DI void block_offsets(BlockCounts& sc, unsigned long long m0, unsigned long long m1, int wave, int lane,
                         int& base0, int& base1, int& total0, int& total1) {
  if (lane == 0) { sc.n[0][wave] = __builtin_popcountll(m0); sc.n[1][wave] = __builtin_popcountll(m1); }
  __syncthreads();
  base0 = base1 = total0 = total1 = 0;
#pragma unroll
  for (int w = 0; w < CULL_WAVES; ++w) {
    if (w < wave) { base0 += sc.n[0][w]; base1 += sc.n[1][w]; }
    total0 += sc.n[0][w]; total1 += sc.n[1][w];
  }
  __syncthreads();     // sc is rewritten by the next call
}

template <bool COUNT>
__global__ __launch_bounds__(CULL_THREADS) void cull_lbvh_block_kernel(const RenderParams p) {
  __shared__ uint32_t buf[CULL_BUF];
  __shared__ int stack[PB_STACK_CAP + 2 * CULL_THREADS];
  __shared__ BlockCounts sc;
  __shared__ uint32_t s_chunk;
  __shared__ int s_any;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int group = blockIdx.x;
  const int nsub = p.ssaa * p.ssaa;
  const int tile = group / nsub, sub = group % nsub;
  const int si = sub / p.ssaa, sj = sub % p.ssaa;
  const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
  bool in_image = false, hit_scene = false;
  Ray r;
  r.active = false; r.maxd = -1.0f; r.dn = v3(0, 0, 1);
  if (wave == 0) {
    // the tile's 64 rays, as in cull_scan_kernel
    const int lx = lane & 7, ly = lane >> 3;
    const int ox = 8 * tx + lx, oy = 8 * tk + ly;
    const int px = p.x0 + ox, py = p.y0 + 8 * tk * p.strip_stride + ly;
    in_image = (ox < p.out_w) && (oy < p.out_h) && (px < p.width) && (py < p.height);
    r = make_ray(p, px * p.ssaa + si, py * p.ssaa + sj);
    r.active = r.active && in_image;
    hit_scene = r.active;
    r.active = r.active && box_gate(p.root.lo, p.root.hi, p.root.cand, p.root.qflags, r.dn, r.maxd);
    if (!r.active) r.maxd = -1.0f;
    float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
    rr[0 * 64 + lane] = r.l.x; rr[1 * 64 + lane] = r.l.y; rr[2 * 64 + lane] = r.l.z;
    rr[3 * 64 + lane] = r.dn.x; rr[4 * 64 + lane] = r.dn.y; rr[5 * 64 + lane] = r.dn.z;
    rr[6 * 64 + lane] = r.maxd; rr[7 * 64 + lane] = 0.0f;
    rr[8 * 64 + lane] = r.d1.x; rr[9 * 64 + lane] = r.d1.y; rr[10 * 64 + lane] = r.d1.z; rr[11 * 64 + lane] = 0.0f;
    const bool any = __ballot(r.active) != 0ull;
    if (lane == 0) { s_any = any ? 1 : 0; stack[0] = 0; }
  }
  __syncthreads();
  int prev = -1;                       // last chunk written for this group (block-uniform)
  if (s_any) {
    V3 pl[4];
    tile_planes(p, tx, tk, pl);
    int top = 1, count = 0;            // block-uniform
    uint32_t next = 0, left = 0, batch = 1;
    // emits buf[from .. from + n) as one chunk
    auto flush = [&](int from, int n) {
      if (left == 0) {
        if (tid == 0) s_chunk = atomicAdd(&p.ctrl[0], batch);
        __syncthreads();
        next = s_chunk;
        left = batch;
        batch *= 2;
      }
      const uint32_t c = next;
      next++; left--;
      if (c < (uint32_t)p.chunk_capacity) {
        uint32_t* dst = p.chunks + (size_t)c * PB_CHUNK_WORDS;
        if (tid == 0) { dst[0] = (uint32_t)group; dst[1] = (uint32_t)n; dst[2] = (uint32_t)prev; dst[3] = 0u; }
        for (int k = tid; k < n; k += CULL_THREADS) dst[4 + k] = buf[from + k];
        prev = (int)c;
      }
      __syncthreads();                 // s_chunk and buf may be rewritten
    };
    while (top > 0) {
      int k = top < CULL_THREADS ? top : CULL_THREADS;
      if (top + k > PB_STACK_CAP) { k = PB_STACK_CAP - top; if (k < 1) k = 1; }   // keep pushes inside the slack
      const int node = (tid < k) ? stack[top - k + tid] : -1;
      top -= k;
      __syncthreads();
      bool in0 = false, in1 = false, lf0 = false, lf1 = false;
      int c0 = 0, c1 = 0;
      uint32_t v0 = 0, v1 = 0;
      if (node >= 0) {
        const LbvhNode nd = p.lbvh_nodes[node];
        c0 = nd.child0; c1 = nd.child1;
        const bool t0 = box_touches_pyramid(nd.lo0, nd.hi0, pl);
        const bool t1 = box_touches_pyramid(nd.lo1, nd.hi1, pl);
        in0 = t0 && c0 >= 0; in1 = t1 && c1 >= 0;
        if (t0 && c0 < 0) { lf0 = axis_touches_pyramid(p.lbvh_sA[~c0], p.lbvh_sB[~c0], pl); if (lf0) v0 = p.lbvh_perm[~c0]; }
        if (t1 && c1 < 0) { lf1 = axis_touches_pyramid(p.lbvh_sA[~c1], p.lbvh_sB[~c1], pl); if (lf1) v1 = p.lbvh_perm[~c1]; }
      }
      int b0, b1, n0, n1;
      unsigned long long m0 = __ballot(in0), m1 = __ballot(in1);
      block_offsets(sc, m0, m1, wave, lane, b0, b1, n0, n1);
      if (in0) stack[top + b0 + (int)lanes_below(m0)] = c0;
      if (in1) stack[top + n0 + b1 + (int)lanes_below(m1)] = c1;
      top += n0 + n1;
      m0 = __ballot(lf0); m1 = __ballot(lf1);
      block_offsets(sc, m0, m1, wave, lane, b0, b1, n0, n1);
      if (lf0) buf[count + b0 + (int)lanes_below(m0)] = v0;
      if (lf1) buf[count + n0 + b1 + (int)lanes_below(m1)] = v1;
      count += n0 + n1;
      __syncthreads();
      int head = 0;
      while (count - head >= PB_CHUNK) { flush(head, PB_CHUNK); head += PB_CHUNK; }
      if (head > 0) {
        const int rest = count - head;           // < PB_CHUNK <= CULL_THREADS
        uint32_t keep = 0;
        if (tid < rest) keep = buf[head + tid];
        __syncthreads();
        if (tid < rest) buf[tid] = keep;
        count = rest;
        __syncthreads();
      }
    }
    if (count > 0) flush(0, count);
    close_unused_chunks(p, group, next, left, tid, CULL_THREADS);
  }
  if (tid == 0) p.group_last[group] = prev;
  if (COUNT && wave == 0) {
    unsigned long long c_gated = 0;
    if (s_any)
      for (int sidx = 0; sidx < p.n_seg; ++sidx) {
        const SegFrame& f = p.frames[sidx];
        c_gated += box_gate(f.lo, f.hi, f.cand, f.qflags, r.dn, r.maxd) ? 1 : 0;
      }
    unsigned long long v[3] = {in_image ? 1ull : 0ull, hit_scene ? 1ull : 0ull, c_gated};
    const int slot[3] = {PB_CNT_RAYS, PB_CNT_RAYS_HIT, PB_CNT_GATED};
    for (int k = 0; k < 3; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&p.counters[slot[k]], x);
    }
  }
}

// Wave-uniform fetch of a segment's record through the scalar unit.  The loads are issued together and
// waited for once: left to itself the compiler sinks every field's load into the branch that uses it,
// which makes three dependent round trips to L2 per candidate.  WHOLE = false fetches the 32 dwords the
// fast evaluation reads.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
template <bool WHOLE>
PB_DI SegFrame fetch_frame(const SegFrame* frames, int sidx) {
  const PB_CONST_AS u32x16* src = (const PB_CONST_AS u32x16*)pb_uniform(frames + sidx);
  SegFrame f;
  u32x16 q0 = src[0], q1 = src[1];
  if (WHOLE) {
    u32x16 q2 = src[2];
    asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2));
    __builtin_memcpy((char*)&f + 128, &q2, 64);
  } else {
    asm volatile("" : "+s"(q0), "+s"(q1));
  }
  __builtin_memcpy((char*)&f, &q0, 64);
  __builtin_memcpy((char*)&f + 64, &q1, 64);
  return f;
}

// Radiance weight times integral for one pair, by either estimator.
// log2_scale: the fast estimator can return its integral times 2^log2_scale for free; FOLDED tells the caller
// whether that happened.
template <bool COUNT, bool PRECISE, int MODE>
PB_DI int eval_pair(const RenderParams& p, int sidx, const SegFrame& f, const Ray& r, double& integral,
                    unsigned long long& c_wrong, float log2_scale = 0.0f) {
  if (MODE == PB_MODE_73HET || MODE == PB_MODE_GENERIC) {
    const HetFrame hf = pb_uniform(p.het)[sidx];
    const PB_CONST_AS float* A = pb_uniform((const float*)p.capA) + 4 * (size_t)sidx;
    const PB_CONST_AS float* B = pb_uniform((const float*)p.capB) + 4 * (size_t)sidx;
    GridView grid;
    grid.z = p.grid_z; grid.side = p.grid_side; grid.fast_div = p.grid_fast_div;
    if (MODE == PB_MODE_GENERIC)
      return eval_pair_generic(f, hf, v3(A[0], A[1], A[2]), v3(B[0], B[1], B[2]), grid, p.one_dimensional != 0, r,
                               p.extinction_c, p.scattering_c, integral);
    return eval_pair_73het(f, hf, v3(A[0], A[1], A[2]), v3(B[0], B[1], B[2]), grid, r, p.extinction_c, p.scattering_c,
                           integral);
  }
  return PRECISE ? eval_pair_721(f, r, p.extinction_c, integral)
                 : eval_pair_721_fast<COUNT>(f, r, p.extinction_c, integral, c_wrong, log2_scale);
}

template <bool COUNT, bool PRECISE, int MODE>
__global__ __launch_bounds__(64) void eval_kernel(const RenderParams p) {
  const int lane = threadIdx.x;
  unsigned long long c_cand = 0, c_cyl = 0, c_valid = 0, c_wrong = 0;
  const uint32_t produced = p.ctrl[0];
  const uint32_t n_chunks = produced < (uint32_t)p.chunk_capacity ? produced : (uint32_t)p.chunk_capacity;
  // chunks are dealt round-robin over the grid (normally one wave per chunk): no shared work counter
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const PB_CONST_AS uint32_t* ch = pb_uniform(p.chunks + (size_t)c * PB_CHUNK_WORDS);
    const int group = (int)ch[0];
    const int n = (int)ch[1];
    if (n == 0) continue;               // an id its group reserved and did not need; in no group's list
    const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
    Ray r;
    r.l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
    r.dn = v3(rr[3 * 64 + lane], rr[4 * 64 + lane], rr[5 * 64 + lane]);
    r.maxd = rr[6 * 64 + lane];
    r.d1 = (MODE != PB_MODE_721) ? v3(rr[8 * 64 + lane], rr[9 * 64 + lane], rr[10 * 64 + lane]) : r.dn;
    r.active = r.maxd >= 0.0f;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < n; ++k) {
      const int sidx = (int)ch[4 + k];
      const SegFrame f = fetch_frame<PRECISE>(p.frames, sidx);
      double integral = 0.0;
      const int res = eval_pair<COUNT, PRECISE, MODE>(p, sidx, f, r, integral, c_wrong);
      if (res == 2) {
        // BeamRadiance.h:455: current += energy * phase_f * rad * _int, product and sum rounded separately
        acc[0] = acc[0] + f.W[0] * integral;
        acc[1] = acc[1] + f.W[1] * integral;
        acc[2] = acc[2] + f.W[2] * integral;
      }
      if (COUNT) { c_cand += r.active ? 1 : 0; c_cyl += (res >= 1) ? 1 : 0; c_valid += (res == 2) ? 1 : 0; }
    }
    double* part = p.partials + (size_t)c * 3 * 64;
    part[0 * 64 + lane] = acc[0]; part[1 * 64 + lane] = acc[1]; part[2 * 64 + lane] = acc[2];
  }
  if (COUNT) {
    unsigned long long v[4] = {c_cand, c_cyl, c_valid, c_wrong};
    const int slot[4] = {PB_CNT_CAND, PB_CNT_CYL, PB_CNT_VALID, PB_CNT_SHORTCUT_WRONG};
    for (int k = 0; k < 4; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&p.counters[slot[k]], x);
    }
  }
}

__global__ __launch_bounds__(64) void reduce_kernel(const RenderParams p) {
  const int lane = threadIdx.x;
  const int tile = blockIdx.x;
  const int nsub = p.ssaa * p.ssaa;
  const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
  const int ox = 8 * tx + (lane & 7), oy = 8 * tk + (lane >> 3);
  double acc[3] = {0.0, 0.0, 0.0};
  for (int sub = 0; sub < nsub; ++sub) {
    int c = p.group_last[tile * nsub + sub];
    while (c >= 0) {
      const double* part = p.partials + (size_t)c * 3 * 64;
      acc[0] = acc[0] + part[0 * 64 + lane];
      acc[1] = acc[1] + part[1 * 64 + lane];
      acc[2] = acc[2] + part[2 * 64 + lane];
      c = (int)p.chunks[(size_t)c * PB_CHUNK_WORDS + 2];
    }
  }
  if (ox < p.out_w && oy < p.out_h) {
    float* o = p.out + ((size_t)oy * p.out_w + ox) * 3;
    const double inv = (double)p.ssaa;        // skeleton.cpp:361: current / (double) SSAA
    o[0] = (float)(acc[0] / inv);
    o[1] = (float)(acc[1] / inv);
    o[2] = (float)(acc[2] / inv);
  }
}


// ---------------------------------------------------------------------------
// Binned frame (cull mode 2).  All camera rays leave (0,0,0), so which tiles a segment can touch
// follows from its projection.
//  ray set-up        one wave per ray group: the 64 rays (ClosestIntersection, root gate), stored.
//  binning           PB_BIN_FAN lanes per segment append it to the list of every macro tile (PB_MACRO x
//                    macro_h tiles) whose pyramid its axis can touch: a counting pass, list offsets, a
//                    filling pass.
//  queue             items = (ray group, slice of its macro tile's list), laid out longest list first.
//  evaluation        one wave per item: lanes first test 64 candidates at a time against the 8x8 tile's
//                    own pyramid and compact the survivors in LDS, then every lane evaluates its ray
//                    against each survivor.
//  finalize_kernel   fixed-point sums -> current / SSAA.
// Small frames (queue built by one block) run these as five launches: setup_count_kernel (ray set-up beside
// the counting pass), queue_build_kernel, fill_items_kernel (filling pass beside the item writing), the
// evaluation kernel, finalize_kernel.  Large or counted frames use one kernel per step and device-wide scans.
// The lists are filled with atomics, hence in no fixed order; the radiance is therefore summed in fixed
// point (PB_FIX_MAGIC), which makes the sum independent of that order and the image reproducible bit
// for bit.
// ---------------------------------------------------------------------------
// The camera rays of a ray group (an 8x8 tile, one sub-sample), one per lane: DrawStandard's per-pixel set-up
// (skeleton.cpp:342-350) and the root gate.  A ray that is outside the image, misses the scene or fails the root gate
// comes back with maxd = -1.
template <class TriPtr, class BoxPtr, class EdgePtr>
PB_DI Ray group_rays(const RenderParams& p, int group, int lane, TriPtr tris, BoxPtr tri_box, EdgePtr tri_edge, bool& hit_scene) {
  const int nsub = p.ssaa * p.ssaa;
  const int tile = group / nsub, sub = group % nsub;
  const int si = sub / p.ssaa, sj = sub % p.ssaa;
  const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
  const int ox = 8 * tx + (lane & 7), oy = 8 * tk + (lane >> 3);
  const int px = p.x0 + ox, py = p.y0 + 8 * tk * p.strip_stride + (lane >> 3);
  const bool in_image = (ox < p.out_w) && (oy < p.out_h) && (px < p.width) && (py < p.height);
  // Triangles whose screen box misses the whole tile are ruled out once per wave (lane i looks at triangle i)
  // instead of once per ray; the per-ray box test and the exact intersection see the rest unchanged.
  unsigned long long tri_mask = ~0ull;
  {
    const float half_w = (p.width * p.ssaa) / (float)2, half_h = (p.height * p.ssaa) / (float)2;
    const float x_lo = (float)((p.x0 + 8 * tx) * p.ssaa + si) - half_w, x_hi = x_lo + (float)(7 * p.ssaa);
    const float y_lo = (float)((p.y0 + 8 * tk * p.strip_stride) * p.ssaa + sj) - half_h, y_hi = y_lo + (float)(7 * p.ssaa);
    bool may = true;
    if (lane < p.n_tri) {
      const float sb_x = tri_box[lane].x, sb_y = tri_box[lane].y, sb_z = tri_box[lane].z, sb_w = tri_box[lane].w;
      const float m = 1e-4f * p.focal;           // the screen boxes are in units of x / z, the rays' z is focal
      may = !(x_hi < sb_x * p.focal - m * (1.0f + fabsf(sb_x)) || x_lo > sb_y * p.focal + m * (1.0f + fabsf(sb_y)) ||
              y_hi < sb_z * p.focal - m * (1.0f + fabsf(sb_z)) || y_lo > sb_w * p.focal + m * (1.0f + fabsf(sb_w)));
      // the tile's rectangle against the triangle's own edges (its screen box is much larger than a wall's half)
      if (tri_edge[12 * lane + 9] != 0.0f) {
        const float inv_f = 1.0f / p.focal;
        const float u_lo = x_lo * inv_f, u_hi = x_hi * inv_f, v_lo = y_lo * inv_f, v_hi = y_hi * inv_f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float a = tri_edge[12 * lane + 3 * k], b = tri_edge[12 * lane + 3 * k + 1], c = tri_edge[12 * lane + 3 * k + 2];
          const float farthest_in = a * (a >= 0.0f ? u_hi : u_lo) + b * (b >= 0.0f ? v_hi : v_lo) + c;
          may = may && !(farthest_in < -2e-3f);
        }
      }
    }
    tri_mask = __ballot(may);
  }
  Ray r = make_ray(p, px * p.ssaa + si, py * p.ssaa + sj, tri_mask, tris, tri_box);
  r.active = r.active && in_image;
  hit_scene = r.active;
  r.active = r.active && box_gate(p.root.lo, p.root.hi, p.root.cand, p.root.qflags, r.dn, r.maxd);
  if (!r.active) r.maxd = -1.0f;
  return r;
}

// tris / tri_box: the prepared scene in global memory, or the copy the launch staged in LDS.
template <bool COUNT, class TriPtr, class BoxPtr, class EdgePtr>
PB_DI void ray_setup_body(const RenderParams& p, int group, int lane, TriPtr tris, BoxPtr tri_box, EdgePtr tri_edge) {
  const int nsub = p.ssaa * p.ssaa;
  const int tile = group / nsub, sub = group % nsub;
  const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
  const int ox = 8 * tx + (lane & 7), oy = 8 * tk + (lane >> 3);
  const int px = p.x0 + ox, py = p.y0 + 8 * tk * p.strip_stride + (lane >> 3);
  const bool in_image = (ox < p.out_w) && (oy < p.out_h) && (px < p.width) && (py < p.height);
  if (p.residue && group == 0) {      // the frame's residue queues start empty
    uint32_t* counters = (uint32_t*)p.residue;
    for (int q = lane; q < PB_RESIDUE_QUEUES; q += 64) counters[q * 32] = 0u;
  }
  if (p.fused_queue) {
    // this launch also clears the pixel sums the frame accumulates into (the bin counters are cleared by
    // the previous frame's finalize_kernel)
    if (sub == 0) {
      unsigned long long* a = p.accum + (size_t)tile * PB_ACC_PLANES * 64 + lane;   // [tile][plane][lane]
      if (p.finalize_in_eval) {
        // no evaluation item will come by this tile: its pixels are written here; a single item writes them itself
        // and needs no sums
        const uint32_t items_of_tile = pb_uniform(p.tile_items)[tile];
        if (items_of_tile > 1u) {
          a[0] = 0ull; a[64] = 0ull; a[128] = 0ull; a[192] = 0ull;
          if (lane == 0) p.tile_done[tile] = 0u;
        } else if (items_of_tile == 0u && ox < p.out_w && oy < p.out_h) {
          float* o = p.out + ((size_t)oy * p.out_w + ox) * 3;
          o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
        }
      } else {
        a[0] = 0ull; a[64] = 0ull; a[128] = 0ull; a[192] = 0ull;
      }
    }
  }
  bool hit_scene;
  Ray r = group_rays(p, group, lane, tris, tri_box, tri_edge, hit_scene);
  float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  rr[0 * 64 + lane] = r.l.x; rr[1 * 64 + lane] = r.l.y; rr[2 * 64 + lane] = r.l.z;
  rr[3 * 64 + lane] = r.dn.x; rr[4 * 64 + lane] = r.dn.y; rr[5 * 64 + lane] = r.dn.z;
  // row 7, lanes 0..11: the four side planes of the tile's pyramid, which every evaluation item of the tile needs
  float plane_word = 0.0f;
  {
    // lane j computes component j % 3 of plane j / 3 with tile_planes' own arithmetic (one normalisation per lane
    // instead of four by all of them)
    const int k = (lane < 12) ? lane / 3 : 0, comp = lane - 3 * (lane / 3);
    const V3 n = tile_plane(p, tx, tk, k);
    plane_word = (lane < 12) ? (comp == 0 ? n.x : (comp == 1 ? n.y : n.z)) : 0.0f;
  }
  rr[6 * 64 + lane] = r.maxd; rr[7 * 64 + lane] = plane_word;
  if (p.store_d1) {    // the once-normalised direction: Integral_73 / Integral_73Het and the cones' Integral_721_ada
    rr[8 * 64 + lane] = r.d1.x; rr[9 * 64 + lane] = r.d1.y; rr[10 * 64 + lane] = r.d1.z;
  }
  const bool any = __ballot(r.active) != 0ull;
  if (lane == 0) p.group_last[group] = any ? 0 : -1;
  if (COUNT) {
    unsigned long long c_gated = 0;
    if (any)
      for (int sidx = 0; sidx < p.n_seg; ++sidx) {
        const SegFrame f = pb_uniform(p.frames)[sidx];
        c_gated += box_gate(f.lo, f.hi, f.cand, f.qflags, r.dn, r.maxd) ? 1 : 0;
      }
    unsigned long long v[3] = {in_image ? 1ull : 0ull, hit_scene ? 1ull : 0ull, c_gated};
    const int slot[3] = {PB_CNT_RAYS, PB_CNT_RAYS_HIT, PB_CNT_GATED};
    for (int k = 0; k < 3; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&p.counters[slot[k]], x);
    }
  }
}

template <bool COUNT>
__global__ __launch_bounds__(64) void ray_setup_kernel(const RenderParams p) {
  ray_setup_body<COUNT>(p, blockIdx.x, threadIdx.x, p.tris, p.tri_box, p.tri_edge);
}

// Inward unit normals of the side planes of the pyramid spanned by the sub-rays of the pixel rectangle
// [px0, px1] x [py0, py1] (pixel coordinates of this launch's frame), widened by half a sub-pixel.
PB_DI void pixel_rect_planes(const RenderParams& p, int px0, int px1, int py0, int py1, V3 pl[4]) {
  const float half_w = (p.width * p.ssaa) / (float)2, half_h = (p.height * p.ssaa) / (float)2;
  const float xl = (float)(px0 * p.ssaa) - half_w - 0.5f;
  const float xr = (float)(px1 * p.ssaa + p.ssaa - 1) - half_w + 0.5f;
  const float yt = (float)(py0 * p.ssaa) - half_h - 0.5f;
  const float yb = (float)(py1 * p.ssaa + p.ssaa - 1) - half_h + 0.5f;
  pl[0] = normalize3(v3(p.focal, 0.0f, -xl));
  pl[1] = normalize3(v3(-p.focal, 0.0f, xr));
  pl[2] = normalize3(v3(0.0f, p.focal, -yt));
  pl[3] = normalize3(v3(0.0f, -p.focal, yb));
}

struct TileRect { int x0, x1, y0, y1; };   // inclusive; empty if x1 < x0

// Macro tiles whose pyramid can come within `pad` of the segment (a, b): bounds of the two projected
// end spheres.  Segments reaching behind z = 0.05 are tested against every macro tile.
PB_DI TileRect macro_rect(const RenderParams& p, V3 a, V3 b, float pad, int mtx, int mty) {
  TileRect r;
  r.x0 = 0; r.x1 = mtx - 1; r.y0 = 0; r.y1 = mty - 1;
  if (fminf(a.z, b.z) - pad <= 0.05f) return r;
  const float half_w = (p.width * p.ssaa) / (float)2, half_h = (p.height * p.ssaa) / (float)2;
  float lo[2] = {3.4e38f, 3.4e38f}, hi[2] = {-3.4e38f, -3.4e38f};
  const V3 pts[2] = {a, b};
  for (int k = 0; k < 2; ++k) {
    const float c[2] = {pts[k].x, pts[k].y};
    const float zn = 1.0f / (pts[k].z - pad), zf = 1.0f / (pts[k].z + pad);
    for (int ax = 0; ax < 2; ++ax) {
      const float m = c[ax] - pad, q = c[ax] + pad;
      lo[ax] = fminf(lo[ax], fminf(m * zn, m * zf));
      hi[ax] = fmaxf(hi[ax], fmaxf(q * zn, q * zf));
    }
  }
  // sub-ray coordinate s = u * focal + half; pixel = s / ssaa; one pixel of margin
  const float inv = 1.0f / (float)p.ssaa;
  const float sx = 1.0f / (float)(8 * p.macro_w), sy = 1.0f / (float)(8 * p.macro_h * p.strip_stride);
  const float fx0 = ((lo[0] * p.focal + half_w) * inv - 1.0f - (float)p.x0) * sx;
  const float fx1 = ((hi[0] * p.focal + half_w) * inv + 1.0f - (float)p.x0) * sx;
  const float fy0 = ((lo[1] * p.focal + half_h) * inv - 1.0f - (float)p.y0) * sy;
  const float fy1 = ((hi[1] * p.focal + half_h) * inv + 1.0f - (float)p.y0) * sy;
  if (fx1 < 0.0f || fy1 < 0.0f) { r.x1 = -1; return r; }
  r.x0 = (int)fmaxf(floorf(fx0), 0.0f);
  r.x1 = (int)fminf(floorf(fx1), (float)(mtx - 1));
  r.y0 = (int)fmaxf(floorf(fy0), 0.0f);
  r.y1 = (int)fminf(floorf(fy1), (float)(mty - 1));
  return r;
}

// Clip the parameter interval [t0, t1] of the axis a-b (widened by `slack`) with the half-space n.x >= 0.
PB_DI bool clip_axis(const V3 n, const float4 A, const float4 B, float slack, float& t0, float& t1) {
  const float fa = n.x * A.x + n.y * A.y + n.z * A.z + slack;
  const float fb = n.x * B.x + n.y * B.y + n.z * B.z + slack;
  if (fa < 0.0f && fb < 0.0f) return false;
  const float t = fa * fast_rcp(fa - fb);          // only used when fa and fb differ in sign
  if (fa < 0.0f && fb >= 0.0f) t0 = fmaxf(t0, t - 1e-5f);
  if (fb < 0.0f && fa >= 0.0f) t1 = fminf(t1, t + 1e-5f);
  return t0 <= t1;
}

// Visits every macro tile the axis of segment s can touch, for the rows my0 + k, my0 + k + PB_BIN_FAN, ...
// of its rectangle: clips the axis with the row's two planes, then with each column's.
template <class F>
PB_DI void for_each_macro_tile(const RenderParams& p, const float4 A4, const float4 B4, int k, int mtx, int mty, F&& hit) {
  const float pad = A4.w * 1.001f + 1e-5f;
  const TileRect rc = macro_rect(p, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), pad, mtx, mty);
  const float half_w = (p.width * p.ssaa) / (float)2, half_h = (p.height * p.ssaa) / (float)2;
  for (int my = rc.y0 + k; my <= rc.y1; my += PB_BIN_FAN) {
    // rows of the macro tile (with interleaved strips its tile rows are strip_stride apart; the pyramid
    // then also spans the rows in between: conservative)
    const int tk0 = my * p.macro_h;
    int tk1 = tk0 + p.macro_h - 1;
    if (tk1 > p.tiles_y - 1) tk1 = p.tiles_y - 1;
    const int py0 = p.y0 + 8 * tk0 * p.strip_stride, py1 = p.y0 + 8 * tk1 * p.strip_stride + 7;
    const float yt = (float)(py0 * p.ssaa) - half_h - 0.5f;
    const float yb = (float)(py1 * p.ssaa + p.ssaa - 1) - half_h + 0.5f;
    float r0 = -1e-5f, r1 = 1.0f + 1e-5f;
    if (!clip_axis(normalize3(v3(0.0f, p.focal, -yt)), A4, B4, pad, r0, r1)) continue;
    if (!clip_axis(normalize3(v3(0.0f, -p.focal, yb)), A4, B4, pad, r0, r1)) continue;
    for (int mx = rc.x0; mx <= rc.x1; ++mx) {
      const int tx0 = mx * p.macro_w;
      int tx1 = tx0 + p.macro_w - 1;
      if (tx1 > p.tiles_x - 1) tx1 = p.tiles_x - 1;
      const int px0 = p.x0 + 8 * tx0, px1 = p.x0 + 8 * tx1 + 7;
      const float xl = (float)(px0 * p.ssaa) - half_w - 0.5f;
      const float xr = (float)(px1 * p.ssaa + p.ssaa - 1) - half_w + 0.5f;
      float t0 = r0, t1 = r1;
      if (!clip_axis(normalize3(v3(p.focal, 0.0f, -xl)), A4, B4, pad, t0, t1)) continue;
      if (!clip_axis(normalize3(v3(-p.focal, 0.0f, xr)), A4, B4, pad, t0, t1)) continue;
      hit(my * mtx + mx);
    }
  }
}

// The same visit, with what it found recorded: bit i of `mask` is the i-th (row, column) pair of this lane's
// share of the rectangle, row by row.  Returns false (mask unusable) when the share has more than 63 pairs.
template <class F>
PB_DI bool for_each_macro_tile_masked(const RenderParams& p, const float4 A4, const float4 B4, int k, int mtx, int mty,
                                      unsigned long long& mask, F&& hit) {
  const float pad = A4.w * 1.001f + 1e-5f;
  const TileRect rc = macro_rect(p, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), pad, mtx, mty);
  const int ncols = rc.x1 - rc.x0 + 1;
  const int nrows = (rc.y1 - (rc.y0 + k)) / PB_BIN_FAN + 1;
  mask = 0ull;
  const bool fits = ncols > 0 && (rc.y0 + k > rc.y1 || nrows * ncols <= 63);
  for_each_macro_tile(p, A4, B4, k, mtx, mty, [&](int m) {
    if (fits) {
      const int my = m / mtx, mx = m % mtx;
      mask |= 1ull << (((my - rc.y0 - k) / PB_BIN_FAN) * ncols + (mx - rc.x0));
    }
    hit(m);
  });
  return fits || ncols <= 0;
}

// Replays a recorded visit without clipping anything.
template <class F>
PB_DI void replay_macro_tiles(const RenderParams& p, const float4 A4, const float4 B4, int k, int mtx, int mty,
                              unsigned long long mask, F&& hit) {
  if (!mask) return;
  const float pad = A4.w * 1.001f + 1e-5f;
  const TileRect rc = macro_rect(p, v3(A4.x, A4.y, A4.z), v3(B4.x, B4.y, B4.z), pad, mtx, mty);
  const int ncols = rc.x1 - rc.x0 + 1;
  while (mask) {
    const int i = __builtin_ctzll(mask);
    mask &= mask - 1;
    hit((rc.y0 + k + (i / ncols) * PB_BIN_FAN) * mtx + rc.x0 + i % ncols);
  }
}

// Binning passes.  A block takes 128 segments, PB_BIN_FAN lanes each (a long beam crosses the whole
// image; one lane per segment walked up to 64 macro tiles one after the other).  Hits are first counted
// per macro tile in LDS, so that a block issues one global atomic per macro tile it touches instead of
// one per hit: every beam passes the tile over the light source, and device-wide atomics on few
// addresses run at about one per 12 ns.  The filling pass reserves a range per (block, macro tile) the
// same way and then walks the tiles a second time to place its segments inside the range.
// LDS = false (more macro tiles than fit in LDS): plain global atomics.
constexpr int BIN_THREADS = 1024;
constexpr int BIN_LDS_MACROS = 4096;      // 2 x 4096 x 4 B = 32 KB of LDS
template <bool FILL, bool LDS>
PB_DI void bin_body(const RenderParams& p, int n_macro, uint32_t* bin_lds, int block) {
  uint32_t* lcount = bin_lds;               // [n_macro] counts, [n_macro] bases
  uint32_t* lbase = bin_lds + n_macro;
  const int mtx = (p.tiles_x + p.macro_w - 1) / p.macro_w, mty = (p.tiles_y + p.macro_h - 1) / p.macro_h;
  const long long g = (long long)block * BIN_THREADS + threadIdx.x;
  const int s = (int)(g / PB_BIN_FAN), k = (int)(g % PB_BIN_FAN);
  const bool live = s < p.n_seg;
  float4 A4 = make_float4(0, 0, 1, 0), B4 = A4;
  if (live) { A4 = p.capA[s]; B4 = p.capB[s]; }
  if (!LDS) {
    if (!live) return;
    for_each_macro_tile(p, A4, B4, k, mtx, mty, [&](int m) {
      if (FILL) {
        const uint32_t at = p.sub_off[m] + atomicAdd(&p.sub_fill[m], 1u);
        if (at < (uint32_t)p.bin_capacity) p.bin_list[at] = (uint32_t)s;
      } else atomicAdd(&p.sub_count[m], 1u);
    });
    return;
  }
  for (int m = threadIdx.x; m < n_macro; m += BIN_THREADS) lcount[m] = 0u;
  __syncthreads();
  // The walk over the macro tiles is done once per frame where possible: the counting pass records what it
  // found per (segment, lane) and the filling pass replays that, twice (block-local counts, then placement).
  constexpr unsigned long long MASK_VALID = 1ull << 63;
  unsigned long long visited = 0ull;
  bool recorded = false;
  auto count_local = [&](int m) { atomicAdd(&lcount[m], 1u); };
  if (live && FILL && p.bin_mask && (p.bin_mask[g] & MASK_VALID)) {
    visited = p.bin_mask[g] & ~MASK_VALID;
    recorded = true;
    replay_macro_tiles(p, A4, B4, k, mtx, mty, visited, count_local);
  } else if (live) {
    recorded = for_each_macro_tile_masked(p, A4, B4, k, mtx, mty, visited, count_local);
    if (!FILL && p.bin_mask) p.bin_mask[g] = recorded ? (visited | MASK_VALID) : 0ull;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < n_macro; m += BIN_THREADS) {
    const uint32_t c = lcount[m];
    if (c) {
      if (FILL) lbase[m] = p.sub_off[m] + atomicAdd(&p.sub_fill[m], c);
      else atomicAdd(&p.sub_count[m], c);
    }
    lcount[m] = 0u;
  }
  if (!FILL) return;
  __syncthreads();
  auto place = [&](int m) {
    const uint32_t at = lbase[m] + atomicAdd(&lcount[m], 1u);
    if (at < (uint32_t)p.bin_capacity) p.bin_list[at] = (uint32_t)s;
  };
  if (live && recorded) replay_macro_tiles(p, A4, B4, k, mtx, mty, visited, place);    // the first walk's findings
  else if (live) for_each_macro_tile(p, A4, B4, k, mtx, mty, place);
}

template <bool FILL, bool LDS>
__global__ __launch_bounds__(BIN_THREADS) void bin_kernel(const RenderParams p, int n_macro) {
  extern __shared__ uint32_t bin_lds[];
  bin_body<FILL, LDS>(p, n_macro, bin_lds, blockIdx.x);
}

// Binning of frames with at most QUEUE_FUSED_MACROS macro tiles: lane = macro tile.  A wave takes `spw`
// consecutive segments (one per lane, handed round with v_readlane) and every lane clips each of them against
// the pyramid of its own macro tile, whose planes it sets up once: no walk over a segment's tiles, no
// dependent memory access per segment, and a segment's incidences with all macro tiles cost one pass of about
// sixty instructions.  What a lane found is a bit mask over the wave's segments, so positions inside a list follow
// from bit counts: per (wave, macro tile) counts go to LDS and one thread per macro tile turns them into the waves'
// offsets inside the block's range.  The count pass stores the block's total per macro tile in blk[block][macro];
// queue_build_kernel turns those into the start of the block's range inside the macro tile's list; the fill pass
// reads that start and writes every lane's segments in ascending order.  No atomics, no counters to clear, and a
// list holds its segments in ascending order whatever the order the blocks ran in.  The fill pass of a frame whose
// offsets are already in place (blk and sub_off kept from the frame that sized the workspace for this segment set
// and frame geometry) needs no count pass before it.
constexpr int MBIN_WAVES = PB_MBIN_WAVES;
constexpr int MBIN_THREADS = 64 * MBIN_WAVES;
constexpr int MBIN_CHUNKS = 4;            // 64 macro tiles per chunk: QUEUE_FUSED_MACROS = 256
PB_DI float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

PB_HDI size_t pb_mbin_lds_bytes(int n_macro) { return sizeof(uint32_t) * (size_t)(MBIN_WAVES + 1) * (size_t)n_macro; }

template <bool FILL>
PB_DI void macro_bin_body(const RenderParams& p, int n_macro, uint32_t* lds, int block) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* lwave = lds;                               // [MBIN_WAVES][n_macro] counts, then offsets inside the block's range
  uint32_t* lbase = lds + MBIN_WAVES * n_macro;        // [n_macro] start of the block's range in bin_list
  const int spw = pb_mbin_spw(p.n_seg);
  const int mtx = (p.tiles_x + p.macro_w - 1) / p.macro_w;
  const int seg0 = (block * MBIN_WAVES + wave) * spw;
  int n_here = p.n_seg - seg0;
  n_here = n_here < 0 ? 0 : (n_here > spw ? spw : n_here);
  float4 A = make_float4(0, 0, 1, 0), B = A;
  if (lane < n_here) { A = p.capA[seg0 + lane]; B = p.capB[seg0 + lane]; }
  uint32_t masks[MBIN_CHUNKS];
#pragma unroll
  for (int c = 0; c < MBIN_CHUNKS; ++c) {
    masks[c] = 0u;
    if (c * 64 < n_macro) {
      const int m_raw = c * 64 + lane;
      const int m = m_raw < n_macro ? m_raw : n_macro - 1;
      const int my = m / mtx, mx = m % mtx;
      // pixel rectangle of macro tile (mx, my); with interleaved strips its tile rows are strip_stride apart and the
      // pyramid also spans the rows in between: conservative
      const int tx0 = mx * p.macro_w, tk0 = my * p.macro_h;
      int tx1 = tx0 + p.macro_w - 1, tk1 = tk0 + p.macro_h - 1;
      if (tx1 > p.tiles_x - 1) tx1 = p.tiles_x - 1;
      if (tk1 > p.tiles_y - 1) tk1 = p.tiles_y - 1;
      V3 pl[4];
      pixel_rect_planes(p, p.x0 + 8 * tx0, p.x0 + 8 * tx1 + 7, p.y0 + 8 * tk0 * p.strip_stride, p.y0 + 8 * tk1 * p.strip_stride + 7, pl);
      uint32_t mask = 0u;
      for (int j = 0; j < n_here; ++j) {
        const float4 a = make_float4(readlane_f(A.x, j), readlane_f(A.y, j), readlane_f(A.z, j), readlane_f(A.w, j));
        const float4 b = make_float4(readlane_f(B.x, j), readlane_f(B.y, j), readlane_f(B.z, j), 0.0f);
        mask |= axis_touches_pyramid(a, b, pl) ? (1u << j) : 0u;
      }
      if (m_raw >= n_macro) mask = 0u;
      masks[c] = mask;
      if (m_raw < n_macro) lwave[wave * n_macro + m_raw] = (uint32_t)__builtin_popcount(mask);
    }
  }
  __syncthreads();
  for (int m = tid; m < n_macro; m += MBIN_THREADS) {
    uint32_t run = 0;
    for (int w = 0; w < MBIN_WAVES; ++w) { const uint32_t c = lwave[w * n_macro + m]; lwave[w * n_macro + m] = run; run += c; }
    if (FILL) lbase[m] = p.sub_off[m] + p.blk[(size_t)block * n_macro + m];
    else p.blk[(size_t)block * n_macro + m] = run;
  }
  if (!FILL) return;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < MBIN_CHUNKS; ++c) {
    uint32_t mask = masks[c];
    if (mask) {
      const int m = c * 64 + lane;
      uint32_t at = lbase[m] + lwave[wave * n_macro + m];
      while (mask) {
        const int j = __builtin_ctz(mask);
        mask &= mask - 1;
        if (at < (uint32_t)p.bin_capacity) p.bin_list[at] = (uint32_t)(seg0 + j);
        ++at;
      }
    }
  }
}

template <bool FILL>
__global__ __launch_bounds__(MBIN_THREADS) void macro_bin_kernel(const RenderParams p, int n_macro) {
  extern __shared__ uint32_t bin_lds[];
  macro_bin_body<FILL>(p, n_macro, bin_lds, blockIdx.x);
}

// Length of every macro tile's list (sum of its sub-counters).
__global__ __launch_bounds__(256) void macro_count_kernel(const RenderParams p, int n_macro) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_macro) return;
  uint32_t cnt = 0;
  for (int k = 0; k < PB_BIN_SUBS; ++k) cnt += p.sub_count[(size_t)m * PB_BIN_SUBS + k];
  p.macro_cnt[m] = cnt;
}

// Rank of every macro tile by list length, longest first (ties by index).  The evaluation queue is laid
// out in this order: the items of the busiest tiles are started first and the short ones fill the gaps,
// instead of a few long items being left running alone at the end of the kernel.
__global__ __launch_bounds__(256) void macro_rank_kernel(const RenderParams p, int n_macro) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_macro) return;
  const uint32_t mine = p.macro_cnt[m];
  uint32_t rank = 0;
  for (int j = 0; j < n_macro; ++j) {
    const uint32_t c = p.macro_cnt[j];
    rank += (c > mine || (c == mine && j < m)) ? 1u : 0u;
  }
  p.macro_rank[m] = rank;
}

// Queue slot of tile t: macro tiles in rank order, PB_MACRO x macro_h slots each.
PB_DI int tile_slot(const RenderParams& p, int t, int& m) {
  const int mtx = (p.tiles_x + p.macro_w - 1) / p.macro_w;
  const int tx = t % p.tiles_x, tk = t / p.tiles_x;
  m = (tk / p.macro_h) * mtx + tx / p.macro_w;
  return (int)p.macro_rank[m] * (p.macro_w * p.macro_h) + (tk % p.macro_h) * p.macro_w + tx % p.macro_w;
}

// Items of a tile: the list of its macro tile cut into slices of p.slice, for each sub-sample group.
// A tile none of whose rays is active gets none.  item_off is indexed by queue slot.
__global__ __launch_bounds__(256) void tile_items_kernel(const RenderParams p, int n_tiles) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const int nsub = p.ssaa * p.ssaa;
  int m;
  const int slot = tile_slot(p, t, m);
  const uint32_t cnt = p.macro_cnt[m];
  p.tile_count[t] = cnt;
  uint32_t n_items = 0;
  for (int q = 0; q < nsub; ++q)
    if (p.group_last[t * nsub + q] >= 0) n_items += pb_item_slices(p, cnt);
  p.item_off[slot] = n_items;
}

PB_DI void make_items_body(const RenderParams& p, int n_tiles, int t) {
  if (t >= n_tiles) return;
  const uint32_t cnt = p.tile_count[t];
  const uint32_t slices = pb_item_slices(p, cnt);
  const uint32_t len = cnt <= (uint32_t)p.slice_whole ? cnt : (uint32_t)p.slice;
  const int nsub = p.ssaa * p.ssaa;
  int m;
  const int slot = tile_slot(p, t, m);
  const uint32_t list0 = p.sub_off[(size_t)m * PB_BIN_SUBS];
  uint32_t it = p.item_off[slot];
  for (int q = 0; q < nsub; ++q) {
    if (p.group_last[t * nsub + q] < 0) continue;
    for (uint32_t c = 0; c < slices; ++c, ++it) {
      if (it >= (uint32_t)p.item_capacity) { if (p.tile_items) p.tile_items[t] = 0xffffffffu; return; }
      uint32_t* dst = p.items + (size_t)it * 4;
      const uint32_t n = (cnt - c * len < len) ? cnt - c * len : len;
      dst[0] = (uint32_t)(t * nsub + q); dst[1] = list0 + c * len; dst[2] = n; dst[3] = 0u;
    }
  }
  if (p.tile_items) p.tile_items[t] = it - p.item_off[slot];
}

__global__ __launch_bounds__(256) void make_items_kernel(const RenderParams p, int n_tiles) {
  make_items_body(p, n_tiles, blockIdx.x * blockDim.x + threadIdx.x);
}

// Small frames are bound by the number of launches, not by their work: the two binning passes each share
// a launch with a step that does not depend on them.  Blocks [0, bin_blocks) bin; the others set up rays,
// one ray group per wave (count pass), or write the evaluation items (fill pass).
constexpr int ITEM_LANES = 64;
// Ray set-up of a block of MBIN_WAVES waves: one ray group per wave (blocks of sixteen waves, one per compute unit,
// left nothing to overlap a block's staging and store drain with: 16.0 -> 13.6 us at four).  The prepared triangles and their screen
// boxes and edges (128 bytes per triangle, adjacent in memory) are first copied to LDS by the whole block in one round trip:
// read one triangle at a time through the scalar unit, a wave paid a dependent trip to L2 for every triangle its
// tile can see.
constexpr int SCENE_LDS_MAX_TRI = 256;
template <int THREADS = MBIN_THREADS>
PB_DI void ray_role(const RenderParams& p, uint32_t* lds, int ray_block, int groups, int scene_in_lds) {
  struct Box4 { float x, y, z, w; };
  typedef __attribute__((address_space(3))) float LdsFloat;
  typedef __attribute__((address_space(3))) Box4 LdsBox4;
  typedef __attribute__((address_space(3))) TriPrep LdsTriPrep;
  const int group = ray_block * (THREADS / 64) + (int)(threadIdx.x >> 6);
  if (scene_in_lds) {
    LdsFloat* dst = (LdsFloat*)lds;
    const float* src = (const float*)p.tris;            // n_tri TriPrep records (16 floats), screen boxes (4), screen edges (12)
    for (int i = threadIdx.x; i < p.n_tri * 32; i += THREADS) dst[i] = src[i];
    __syncthreads();
    if (group < groups)
      ray_setup_body<false>(p, group, threadIdx.x & 63, (const LdsTriPrep*)lds, (const LdsBox4*)lds + 4 * p.n_tri,
                            (const LdsFloat*)lds + 20 * p.n_tri);
  } else if (group < groups) {
    ray_setup_body<false>(p, group, threadIdx.x & 63, p.tris, p.tri_box, p.tri_edge);
  }
}
__global__ __launch_bounds__(MBIN_THREADS) void setup_count_kernel(const RenderParams p, int n_macro, int bin_blocks, int groups, int scene_in_lds) {
  extern __shared__ __align__(16) uint32_t bin_lds[];
  if ((int)blockIdx.x < bin_blocks) { macro_bin_body<false>(p, n_macro, bin_lds, blockIdx.x); return; }
  ray_role(p, bin_lds, (int)blockIdx.x - bin_blocks, groups, scene_in_lds);
}

// A frame whose list offsets and evaluation items are in place (kept from the frame that sized the workspace for this
// segment set and frame geometry): ray set-up beside the one binning pass it needs, the fill.
__global__ __launch_bounds__(MBIN_THREADS) void setup_fill_kernel(const RenderParams p, int n_macro, int bin_blocks, int groups, int scene_in_lds) {
  extern __shared__ __align__(16) uint32_t bin_lds[];
  if ((int)blockIdx.x < bin_blocks) { if (!(p.debug & 64)) macro_bin_body<true>(p, n_macro, bin_lds, blockIdx.x); return; }
  if (p.debug & 32) return;
  ray_role(p, bin_lds, (int)blockIdx.x - bin_blocks, groups, scene_in_lds);
}

// Ray set-up alone, for frames whose lists are in place.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void setup_rays_kernel(const RenderParams p, int groups, int scene_in_lds) {
  extern __shared__ __align__(16) uint32_t bin_lds[];
  ray_role<THREADS>(p, bin_lds, (int)blockIdx.x, groups, scene_in_lds);
}

__global__ __launch_bounds__(MBIN_THREADS) void fill_items_kernel(const RenderParams p, int n_macro, int bin_blocks, int n_tiles) {
  extern __shared__ uint32_t bin_lds[];
  if ((int)blockIdx.x < bin_blocks) { macro_bin_body<true>(p, n_macro, bin_lds, blockIdx.x); return; }
  // one wave per block writes items: spread over many compute units, the scattered 16-byte stores of four
  // full blocks took as long as the whole binning pass
  if (threadIdx.x < ITEM_LANES) make_items_body(p, n_tiles, ((int)blockIdx.x - bin_blocks) * ITEM_LANES + (int)threadIdx.x);
}

// The steps between the two binning passes in one block (frames of up to QUEUE_FUSED_MACROS macro tiles):
// list offsets of the macro tiles, their rank by list length, items per queue slot and their prefix sum.  Small frames spend more time between launches than in these steps.
constexpr int QUEUE_FUSED_MACROS = 256;            // macro tiles the single-block queue build holds in LDS
constexpr int QUEUE_THREADS = 1024;
__global__ __launch_bounds__(QUEUE_THREADS) void queue_build_kernel(const RenderParams p, int n_macro, int bin_blocks) {
  __shared__ uint32_t s_cnt[256], s_inv[256], s_part[QUEUE_THREADS / 64], s_tot[QUEUE_THREADS];
  const int t = threadIdx.x;
  const int nsub = p.ssaa * p.ssaa;
  const int mtx = (p.tiles_x + p.macro_w - 1) / p.macro_w;
  {
    // List length of every macro tile = sum of the binning blocks' counts, each count replaced by the sum of those
    // before it.  The blocks are split over as many thread groups as the block has room for (a frame of 64 macro tiles
    // and 1849 binning blocks kept 64 threads busy for 0.42 ms, one dependent load and store after the other): a group
    // sums its share (loads only, eight in flight), the groups' sums are scanned in LDS, and a second walk writes the
    // prefixes.
    const int n_groups = QUEUE_THREADS / n_macro;          // >= 4: n_macro <= QUEUE_FUSED_MACROS
    const int m = t % n_macro, g = t / n_macro;
    const int per = (bin_blocks + n_groups - 1) / n_groups;
    const int b0 = g * per, b1 = (b0 + per < bin_blocks) ? b0 + per : bin_blocks;
    const bool works = g < n_groups;
    uint32_t sum = 0;
    if (works) {
      int b = b0;
      for (; b + 8 <= b1; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = p.blk[(size_t)(b + j) * n_macro + m];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += c[j];
      }
      for (; b < b1; ++b) sum += p.blk[(size_t)b * n_macro + m];
      s_tot[g * n_macro + m] = sum;
    }
    __syncthreads();
    if (works) {
      uint32_t run = 0;
      for (int gg = 0; gg < g; ++gg) run += s_tot[gg * n_macro + m];
      if (g == 0) {
        uint32_t total = 0;
        for (int gg = 0; gg < n_groups; ++gg) total += s_tot[gg * n_macro + m];
        s_cnt[m] = total;
      }
      int b = b0;
      for (; b + 8 <= b1; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = p.blk[(size_t)(b + j) * n_macro + m];
#pragma unroll
        for (int j = 0; j < 8; ++j) { p.blk[(size_t)(b + j) * n_macro + m] = run; run += c[j]; }
      }
      for (; b < b1; ++b) {
        uint32_t* at = p.blk + (size_t)b * n_macro + m;
        const uint32_t c = *at;
        *at = run;
        run += c;
      }
    }
  }
  __syncthreads();
  if (t < n_macro) {
    const uint32_t mine = s_cnt[t];
    uint32_t off = 0, rank = 0;
    for (int j = 0; j < n_macro; ++j) {
      const uint32_t c = s_cnt[j];
      off += (j < t) ? c : 0u;
      rank += (c > mine || (c == mine && j < t)) ? 1u : 0u;
    }
    s_inv[rank] = (uint32_t)t;
    p.sub_off[t] = off; p.macro_cnt[t] = mine; p.macro_rank[t] = rank;
    if (t == n_macro - 1) p.ctrl[2] = off + mine;
  }
  __syncthreads();
  const int per_macro = p.macro_w * p.macro_h;
  const int slots = n_macro * per_macro;
  const int per = (slots + QUEUE_THREADS - 1) / QUEUE_THREADS;
  const int slot0 = t * per;
  // tile, list length and number of active groups of a slot; tile < 0: no tile there
  auto slot_info = [&](int slot, uint32_t& cnt, uint32_t& m) -> int {
    m = s_inv[slot / per_macro];
    const int local = slot % per_macro;
    const int tx = (int)(m % mtx) * p.macro_w + local % p.macro_w, tk = (int)(m / mtx) * p.macro_h + local / p.macro_w;
    cnt = s_cnt[m];
    return (tx < p.tiles_x && tk < p.tiles_y) ? tk * p.tiles_x + tx : -1;
  };
  uint32_t mine_items = 0;
  for (int slot = slot0; slot < slot0 + per && slot < slots; ++slot) {
    uint32_t cnt, m;
    const int tile = slot_info(slot, cnt, m);
    if (tile < 0) continue;
    const uint32_t slices = pb_item_slices(p, cnt);
    for (int q = 0; q < nsub; ++q)
      if (p.group_last[tile * nsub + q] >= 0) mine_items += slices;
  }
  // block-wide exclusive scan of mine_items: in-wave shuffles, then the 16 wave totals
  uint32_t incl = mine_items;
  const int lane = t & 63, wave = t >> 6;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int w = 0; w < QUEUE_THREADS / 64; ++w) { const uint32_t v = s_part[w]; s_part[w] = run; run += v; }
    p.ctrl[0] = run;
  }
  __syncthreads();
  uint32_t it = s_part[wave] + incl - mine_items;
  // first item of every slot; make_items_kernel writes the items
  for (int slot = slot0; slot < slot0 + per && slot < slots; ++slot) {
    uint32_t cnt, m;
    const int tile = slot_info(slot, cnt, m);
    p.item_off[slot] = it;
    if (tile < 0) continue;
    p.tile_count[tile] = cnt;
    const uint32_t slices = pb_item_slices(p, cnt);
    for (int q = 0; q < nsub; ++q)
      if (p.group_last[tile * nsub + q] >= 0) it += slices;
  }
}

#ifndef REFINE_UNROLL
#define REFINE_UNROLL 1
#endif
template <bool COUNT, bool PRECISE, int MODE>
PB_DI void eval_bins_body(const RenderParams& p, uint32_t* surv) {
  const int lane = threadIdx.x;
  unsigned long long c_cand = 0, c_cyl = 0, c_valid = 0, c_wrong = 0;
  const uint32_t produced = p.kept_queue ? (uint32_t)p.n_items_host : p.ctrl[0];   // a kept queue has been cut anew since it was built
  const uint32_t n_items = produced < (uint32_t)p.item_capacity ? produced : (uint32_t)p.item_capacity;
  const int nsub = p.ssaa * p.ssaa;
  const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
  if (COUNT && lane == 0) atomicMax(&p.counters[PB_CNT_T_START_INV], ~(unsigned long long)wall_clock64());
  // Items are dealt round-robin: wave w takes items w, w + grid, w + 2 grid, ...  The queue is sorted by
  // weight (longest lists first), so every wave gets a similar mix; a shared work counter costs about 12 ns
  // per item on this chip (all waves hit one address) and was most of the kernel time.
  for (uint32_t c = blockIdx.x; c < n_items; c += gridDim.x) {
    if (COUNT && lane == 0) atomicAdd(&p.counters[PB_CNT_ITEMS], 1ull);
    const PB_CONST_AS uint32_t* it = pb_uniform(p.items + (size_t)c * 4);
    const int group = (int)it[0];
    const uint32_t start = it[1];
    const int n = (int)it[2];
    const int tile = group / nsub;
    const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
    V3 pl[4];
    tile_planes(p, tx, tk, pl);
    const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
    Ray r;
    r.l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
    r.dn = v3(rr[3 * 64 + lane], rr[4 * 64 + lane], rr[5 * 64 + lane]);
    r.maxd = rr[6 * 64 + lane];
    r.d1 = (MODE != PB_MODE_721) ? v3(rr[8 * 64 + lane], rr[9 * 64 + lane], rr[10 * 64 + lane]) : r.dn;
    r.active = r.maxd >= 0.0f;
    unsigned long long ai[3] = {0ull, 0ull, 0ull};
    unsigned long long ag = 0ull;      // contributions of segments whose three weights are equal
    uint32_t nc = 0;
    {
      const int cbase = 0, cn = n;             // n <= PB_SLICE, the capacity of surv
      // 1. lanes test the candidates against this tile's pyramid, REFINE_UNROLL x 64 at a time;
      //    survivors are compacted into LDS
      int ns = 0;
      if (it[3]) {                              // a kept queue whose items hold survivors already
        for (int base = 0; base < cn; base += 64) { const int k = base + lane; if (k < cn) surv[k] = p.surv_list[start + k]; }
        ns = cn;
      }
      else for (int base = 0; base < cn; base += 64 * REFINE_UNROLL) {
        uint32_t cand[REFINE_UNROLL];
        float4 ca[REFINE_UNROLL], cb[REFINE_UNROLL];
#pragma unroll
        for (int j = 0; j < REFINE_UNROLL; ++j) {
          const int k = base + j * 64 + lane;
          cand[j] = (k < cn) ? p.bin_list[start + cbase + k] : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < REFINE_UNROLL; ++j)
          if (cand[j] != 0xffffffffu) { ca[j] = p.capA[cand[j]]; cb[j] = p.capB[cand[j]]; }
#pragma unroll
        for (int j = 0; j < REFINE_UNROLL; ++j) {
          const bool touch = (cand[j] != 0xffffffffu) && axis_touches_pyramid(ca[j], cb[j], pl);
          const unsigned long long mask = __ballot(touch);
          if (touch) surv[ns + (int)lanes_below(mask)] = cand[j];
          ns += __builtin_popcountll(mask);
        }
      }
      __syncthreads();
      // 2. every lane evaluates its own ray against each survivor (wave-uniform record, scalar loads)
      int sidx = ns > 0 ? __builtin_amdgcn_readfirstlane((int)surv[0]) : 0;
      for (int k = 0; k < ns; ++k) {
        const uint32_t next_idx = surv[(k + 1 < ns) ? k + 1 : k];   // read now, needed at the end of the iteration
        const SegFrame f = fetch_frame<PRECISE>(p.frames, sidx);
        double integral = 0.0;
        constexpr bool FOLDED = !PRECISE && MODE == PB_MODE_721;      // the 2^-10 of the fixed-point sums
        const int res = eval_pair<COUNT, PRECISE, MODE>(p, sidx, f, r, integral, c_wrong, FOLDED ? -10.0f : 0.0f);
        if (res == 2) {
          if (MODE != PB_MODE_721) {
            // 1/sin(theta) is unbounded for a beam seen end-on, 1/r^2 at a cone's apex: saturate far above
            // the display range
            integral = fmin(integral, PB_FIX_LIMIT / fmax(f.W[0], fmax(f.W[1], f.W[2])));
          } else {
            // a thin beam's weight grows as 1/r^2: saturate likewise; a NaN integral (degenerate record) adds nothing
            const double cap = (double)f.vcap10 * (FOLDED ? 1.0 : 1024.0);
            integral = (integral >= 0.0) ? fmin(integral, cap) : 0.0;
          }
          if (!FOLDED) integral = integral * PB_FIX_PRESCALE;
          if (f.qflags & 64u) {
            // the three weights are equal (a beam that has not met a coloured wall): one sum serves all channels
            ag += (unsigned long long)__double_as_longlong(__builtin_fma(f.W[0], integral, PB_FIX_MAGIC));
          } else {
            ai[0] += (unsigned long long)__double_as_longlong(__builtin_fma(f.W[0], integral, PB_FIX_MAGIC));
            ai[1] += (unsigned long long)__double_as_longlong(__builtin_fma(f.W[1], integral, PB_FIX_MAGIC));
            ai[2] += (unsigned long long)__double_as_longlong(__builtin_fma(f.W[2], integral, PB_FIX_MAGIC));
          }
          nc++;
        }
        if (COUNT) { c_cand += r.active ? 1 : 0; c_cyl += (res >= 1) ? 1 : 0; c_valid += (res == 2) ? 1 : 0; }
        sidx = __builtin_amdgcn_readfirstlane((int)next_idx);
      }
      __syncthreads();   // surv is rewritten by the next item
    }
    if (nc) {
      // sums are kept per tile, [tile][channel][lane]: one atomic instruction covers 512 contiguous bytes
      // (the adds run at the memory side in 64-byte requests; pixel-major rows cost three times as many)
      unsigned long long* dst = p.accum + (size_t)tile * PB_ACC_PLANES * 64 + lane;
      const unsigned long long bias = (unsigned long long)nc * magic_bits;
      atomicAdd(&dst[0], ai[0] + ag - bias);
      atomicAdd(&dst[64], ai[1] + ag - bias);
      atomicAdd(&dst[128], ai[2] + ag - bias);
    }
  }
  if (COUNT) {
    if (lane == 0) atomicMax(&p.counters[PB_CNT_T_DRAIN_INV], ~(unsigned long long)wall_clock64());
    unsigned long long v[4] = {c_cand, c_cyl, c_valid, c_wrong};
    const int slot[4] = {PB_CNT_CAND, PB_CNT_CYL, PB_CNT_VALID, PB_CNT_SHORTCUT_WRONG};
    for (int k = 0; k < 4; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&p.counters[slot[k]], x);
    }
    if (lane == 0) atomicMax(&p.counters[PB_CNT_T_END], (unsigned long long)wall_clock64());
  }
}

template <bool COUNT, bool PRECISE, int MODE>
__global__ __launch_bounds__(64) void eval_bins_kernel(const RenderParams p) {
  __shared__ uint32_t surv[PB_SLICE];
  eval_bins_body<COUNT, PRECISE, MODE>(p, surv);
}

// The default estimator at eight waves per SIMD (64 VGPRs, two of them spilled): the pair loop waits on
// the scalar loads of the next record, and resident waves are what hides that.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void eval_bins_fast_kernel(const RenderParams p) {
  __shared__ uint32_t surv[PB_SLICE];
  eval_bins_body<false, false, PB_MODE_721>(p, surv);
}

// The default estimator again, one wave per item, with the record loads of the pair loop taken off its critical path.
// eval_bins_fast_kernel fetches a candidate's record and waits for it before the first instruction that uses it: a
// round trip to L2 per candidate, hidden only by other waves (rocprofv3: 36 % of the wave cycles parked on that wait).
// Here the loop keeps two loads in flight while it computes: the half of the current record that only the end of the
// iteration reads (radiance weights, gate box) and the first half of the next candidate's record (everything
// HitCylinder and Integral_721 need).  Scalar loads return out of order, so the one wait per iteration is for both;
// it sits behind the geometry of the current pair.  The loads and the wait are written as inline assembly: the
// compiler would otherwise sink a load to its first use, and it does not see inline-assembly loads when it places
// its own waits, which only makes those waits stricter.  The survivors' indices sit in a VGPR (lane j: survivor j)
// and are picked with v_readlane, so the loop touches LDS once per 64 candidates.  The tile's pyramid planes come
// from row 7 of its ray block (ray_setup_body) instead of being normalised again by every item.
// Tried and dropped: the pair loop cut behind the discriminant, the crossing lanes (half of them on average) compacted
// through an LDS queue and the rest of the pair evaluated 64 pairs to a wave, each lane fetching its segment's fields with
// 16-byte loads and its ray with ds_bpermute (the shape eval_items_het2_kernel has): frames bit for bit the same, and
// 0.0367 -> 0.0506 ms at 1000 photons, 0.177 -> 0.279 ms at 10 000, 3.15 -> 6.02 ms at 100 000 / 1024^2.  With one pair in two
// crossing there is a drain every two survivors, and a drain's five per-lane gathers of 64 different records cost the
// vector memory pipe more cycles than the half-empty lanes cost the vector ALU.
// Also: the survivors' records copied to LDS and read back as vector operands (a vector instruction
// with a scalar-register source issues in 4.3 cycles against 2.6 on vector registers alone, tools/micro/valu_rate.hip):
// 0.245 ms against 0.223 ms at 10 000 photons, 0.277 ms with the next record requested a candidate ahead (80 registers).
struct SegHalfA { float Rinv[9]; float ox2, oz2, c4, oy, len, tb0; uint32_t qflags; };      // SegFrame dwords 0..15
struct SegHalfB { double W[3]; float lo[3], hi[3], cand[3]; float vcap10; };                  // SegFrame dwords 16..31
static_assert(sizeof(SegHalfA) == 64 && sizeof(SegHalfB) == 64, "halves of the first 128 bytes of SegFrame");

PB_DI void smem_load16(u32x16& dst, const SegFrame* base, uint32_t byte_offset) {
  asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(dst) : "s"(base), "s"(byte_offset));
}
PB_DI void smem_wait(u32x16& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
// The compiler takes a register written by smem_load16 to hold its value from that statement on and to be free after
// the last statement that names it, while the data arrives whenever the load completes: every path from a load has to
// reach a wait that NAMES the destination, or the registers can be given to something else (a mask, a restored pointer)
// with the load still in flight and the data lands on top of it.  tools/check_smem_inflight.py checks the assembly.
//  - smem_wait_keep: a wait on a path only some lanes take.  A scalar register redefined under a divergent branch
//    cannot be named after the branches join, so this one leaves `a` as it is and is tied to the two per-lane values
//    everything downstream of it is computed from instead (what reads `a` there also reads one of them, or sits in a
//    block entered on one of them).
//  - smem_wait_both(a, kept): the wait every path reaches; names `kept` as well, so that its registers stay reserved on
//    the paths that never looked at it.
PB_DI void smem_wait_keep(const u32x16& a, float& x, float& y) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y) : "s"(a)); }
PB_DI void smem_wait_both(u32x16& a, const u32x16& kept) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a) : "s"(kept)); }

struct PairSums { unsigned long long ai[3]; unsigned long long ag; uint32_t nc, ng; };   // nc / ng: terms in ai / in ag

// One candidate: `cur` holds the first half of its record (loaded and waited for); requests the second half and the
// first half of the next candidate's record into `nxt`, evaluates, waits, accumulates.
PB_DI void pair_step(const SegFrame* frames, u32x16& cur, u32x16& nxt, uint32_t off_cur, uint32_t off_nxt, const Ray& r,
                     float extinction, PairSums& sum) {
  u32x16 second;
  smem_load16(second, frames, off_cur + 64u);
  smem_load16(nxt, frames, off_nxt);
  SegHalfA a;
  __builtin_memcpy(&a, &cur, 64);
  float qa, qb;
  const float b24ac = pair_721_fast_quadratic(a.Rinv, a.ox2, a.oz2, a.c4, r.l, qa, qb);
  if (!(b24ac < 0)) {
    // what is done with a valid crossing sits at the innermost point of the pair's branches: nothing it needs has to
    // be given a value on the paths that leave early
    pair_721_fast_crossing_then(qa, qb, b24ac, v3(a.Rinv[1], a.Rinv[4], a.Rinv[7]), a.oy, a.len, a.tb0, r.l, extinction, -10.0f,
                                [&](float tc_minus, float value) __attribute__((always_inline)) {
      smem_wait_keep(second, tc_minus, value);
      SegHalfB b;
      __builtin_memcpy(&b, &second, 64);
      bool ok = true;
      if (!pair_721_fast_sure(a.qflags, tc_minus, r.maxd)) ok = pair_721_fast_gate(b.lo, b.hi, b.cand, a.qflags, r);
      if (ok) {
        // carries the 2^-10 of the fixed-point sums; the median saturates a thin beam's 1/r^2 weight far above the
        // display range and turns a NaN (degenerate record) into no contribution
        const double integral = (double)__builtin_amdgcn_fmed3f(value, 0.0f, b.vcap10);
        // Every term's first channel goes to the sum all channels share; a segment whose weights differ (its beam has
        // met a coloured wall) adds, for the other two channels, what they have over the first - differences of two
        // anchored terms, so they carry no anchor and need no count.  Integer arithmetic: the channels' sums are the
        // same bits as with one sum per channel.  (One-sided: the common path joins nothing, so no value is moved.)
        const unsigned long long first = (unsigned long long)__double_as_longlong(__builtin_fma(b.W[0], integral, PB_FIX_MAGIC));
        sum.ag += first;
        sum.ng++;
        if (!(a.qflags & 64u)) {
          sum.ai[1] += (unsigned long long)__double_as_longlong(__builtin_fma(b.W[1], integral, PB_FIX_MAGIC)) - first;
          sum.ai[2] += (unsigned long long)__double_as_longlong(__builtin_fma(b.W[2], integral, PB_FIX_MAGIC)) - first;
        }
      }
    });
  }
  smem_wait_both(nxt, second);  // on every path (both loads have landed then): the next step reads `nxt`
}

// What an evaluation item does with its lane sums: adds them to its tile's accumulators and, where the frame has no
// write-out launch of its own, writes the tile's pixels if it is the last of the tile's items to arrive.
template <bool FINALIZE>
PB_DI void item_writeout(const RenderParams& p, int tile, int lane, const PairSums& sum) {
  const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
  unsigned long long* dst = p.accum + (size_t)tile * PB_ACC_PLANES * 64 + lane;
  // A segment whose beam has met no coloured wall weighs the three channels alike (most do): its terms go to the
  // tile's fourth plane, one atomic instruction instead of three; the pixel write-out adds that plane to each channel.
  const unsigned long long grey = sum.ag - (unsigned long long)sum.ng * magic_bits;
  const unsigned long long bias = (unsigned long long)sum.nc * magic_bits;
  if (!FINALIZE) {
    if (sum.ng) atomicAdd(&dst[192], grey);
    if (sum.nc) {
      atomicAdd(&dst[0], sum.ai[0] - bias);
      atomicAdd(&dst[64], sum.ai[1] - bias);
      atomicAdd(&dst[128], sum.ai[2] - bias);
    }
  } else {
    // The item that finds all the others of its tile done writes the tile's pixels.  Everything is exchanged through
    // atomic read-modify-writes, which execute at the memory side whatever any cache holds, so no cache has to be
    // written back or invalidated (an agent-scope release per item made this kernel fifteen times slower): the
    // additions return their old values, the wave waits for them - the sums are then in place - and only then counts
    // itself done; the wave that reads the full count fetches the sums by adding zero.
    const uint32_t items_of_tile = pb_uniform(p.tile_items)[tile];
    if (items_of_tile == 1u) {
      // the tile's only item: its sums are the pixels, nothing to exchange
      const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
      const int ox = 8 * tx + (lane & 7), oy = 8 * tk + (lane >> 3);
      if (ox < p.out_w && oy < p.out_h) {
        const double inv = PB_FIX_SCALE / (double)p.ssaa;      // skeleton.cpp:361: current / (double) SSAA
        float* o = p.out + ((size_t)oy * p.out_w + ox) * 3;
        o[0] = (float)((double)(long long)(sum.ai[0] - bias + grey) * inv);
        o[1] = (float)((double)(long long)(sum.ai[1] - bias + grey) * inv);
        o[2] = (float)((double)(long long)(sum.ai[2] - bias + grey) * inv);
      }
      return;
    }
    unsigned long long r0 = 0ull, r1 = 0ull, r2 = 0ull, r3 = 0ull;
    if (sum.ng) r3 = atomicAdd(&dst[192], grey);
    if (sum.nc) {
      r0 = atomicAdd(&dst[0], sum.ai[0] - bias);
      r1 = atomicAdd(&dst[64], sum.ai[1] - bias);
      r2 = atomicAdd(&dst[128], sum.ai[2] - bias);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
    uint32_t before = 0u;
    if (lane == 0) before = __hip_atomic_fetch_add(&p.tile_done[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    if (before + 1u == items_of_tile) {
      asm volatile("" : : : "memory");
      // fetched by exchange with zero: the sums and the arrival counter are left as the next frame needs them, so a
      // frame whose items set up their own rays (no launch before the evaluation that could clear them) can follow
      const unsigned long long vg = atomicExch(&dst[192], 0ull);
      const unsigned long long v0 = atomicExch(&dst[0], 0ull) + vg, v1 = atomicExch(&dst[64], 0ull) + vg, v2 = atomicExch(&dst[128], 0ull) + vg;
      if (lane == 0) __hip_atomic_store(&p.tile_done[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int tx = tile % p.tiles_x, tk = tile / p.tiles_x;
      const int ox = 8 * tx + (lane & 7), oy = 8 * tk + (lane >> 3);
      if (ox < p.out_w && oy < p.out_h) {
        const double inv = PB_FIX_SCALE / (double)p.ssaa;      // skeleton.cpp:361: current / (double) SSAA
        float* o = p.out + ((size_t)oy * p.out_w + ox) * 3;
        o[0] = (float)((double)(long long)v0 * inv);
        o[1] = (float)((double)(long long)v1 * inv);
        o[2] = (float)((double)(long long)v2 * inv);
      }
    }
  }
}

// INLINE (kept queues of survivors only): the frame is this one launch.  Every item sets up its tile's rays itself
// (group_rays, the scene read through the scalar and vector caches) instead of reading them from a ray block an earlier
// launch wrote; the blocks behind the items write the black pixels of the tiles that have no item (one lane per tile
// looks, the wave writes).  The sums of tiles with several items and their arrival counters are zero when the frame
// starts because the last arriver of the frame before left them so.
template <bool FINALIZE, bool INLINE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void eval_items_fast_kernel(const RenderParams p) {
  __shared__ uint32_t surv[INLINE ? 1 : PB_SLICE];
  const int lane = threadIdx.x;
  if (INLINE && (int)blockIdx.x >= p.inline_item_blocks) {
    const int n_tiles = p.tiles_x * p.tiles_y;
    const int t = ((int)blockIdx.x - p.inline_item_blocks) * 64 + lane;
    unsigned long long empty = __ballot(t < n_tiles && p.tile_items[t < n_tiles ? t : 0] == 0u);
    while (empty) {
      const int j = __builtin_ctzll(empty);
      empty &= empty - 1ull;
      const int tile = ((int)blockIdx.x - p.inline_item_blocks) * 64 + j;
      const int ox = 8 * (tile % p.tiles_x) + (lane & 7), oy = 8 * (tile / p.tiles_x) + (lane >> 3);
      if (ox < p.out_w && oy < p.out_h) {
        float* o = p.out + ((size_t)oy * p.out_w + ox) * 3;
        o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
      }
    }
    return;
  }
  // the number of items is the one the host measured when it sized the workspace for this segment set and frame
  // (the queue build leaves the same number in ctrl[0]): an argument instead of a trip to memory before the first load
  const uint32_t produced = (uint32_t)p.n_items_host;
  const uint32_t n_items = produced < (uint32_t)p.item_capacity ? produced : (uint32_t)p.item_capacity;
  // Workgroups go round the eight XCDs by their index, and each XCD has an L2 of its own: the items of one tile, which
  // are neighbours in the queue and read the same rays and much the same records, are kept on one XCD by handing out
  // the queue in runs of xcd_run items (the runs still go round the XCDs, longest lists first).
  uint32_t c = blockIdx.x;
  if (p.xcd_run > 0) {
    const uint32_t run_len = (uint32_t)p.xcd_run, xcd = c & 7u, j = c >> 3;
    c = ((j / run_len) * 8u + xcd) * run_len + j % run_len;
  }
  if (c >= n_items) return;
  const PB_CONST_AS uint32_t* it = pb_uniform(p.items + (size_t)c * 4);
  const int group = (int)it[0];
  const uint32_t start = it[1];
  const int cn = (int)it[2];
  const int tile = group / (p.ssaa * p.ssaa);
  const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  const uint32_t refined = INLINE ? 1u : it[3];
  Ray r;
  if (INLINE) {
    bool hit_scene;
    r = group_rays(p, group, lane, p.tris, p.tri_box, p.tri_edge, hit_scene);
  } else {
    r.l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
    r.dn = v3(rr[3 * 64 + lane], rr[4 * 64 + lane], rr[5 * 64 + lane]);
    r.maxd = rr[6 * 64 + lane];
    r.d1 = r.dn;
    r.active = r.maxd >= 0.0f;
  }
  // 1. lanes test the candidates against this tile's pyramid, 64 at a time; survivors are compacted into LDS - unless
  //    the item is one of a kept queue and holds survivors already: those are read where they lie
  int ns = cn;
  const uint32_t* kept = p.surv_list + start;
  if (!refined) {
    V3 pl[4];
    const PB_CONST_AS float* w = pb_uniform(rr + 7 * 64);
    pl[0] = v3(w[0], w[1], w[2]); pl[1] = v3(w[3], w[4], w[5]); pl[2] = v3(w[6], w[7], w[8]); pl[3] = v3(w[9], w[10], w[11]);
    ns = refine_candidates<false>(p, start, cn, pl, surv, lane, 0u);
    __syncthreads();
  }
  // A SIMD shares its issue slots among up to eight items, and the kernel ends with its longest one: items with many
  // survivors go first in the SIMD's arbitration, the short ones fill the slots they leave.
  if (ns > 32) __builtin_amdgcn_s_setprio(3);
  else if (ns > 16) __builtin_amdgcn_s_setprio(2);
  else if (ns > 8) __builtin_amdgcn_s_setprio(1);
  // 2. every lane evaluates its own ray against each survivor
  PairSums sum;
  sum.ai[0] = sum.ai[1] = sum.ai[2] = sum.ag = 0ull; sum.nc = sum.ng = 0u;
  const SegFrame* frames = p.frames;
  // kept in a vector register: the pair loop's SGPRs are taken by the record buffers (as a scalar it was spilled to a
  // VGPR lane and read back with v_readlane for every valid pair), and operands from vector registers issue faster
  float extinction = p.extinction_c;
  asm volatile("" : "+v"(extinction));
  if (p.debug & 128) ns = 0;             // timing experiment: no pair loop
  if (p.debug & 256) return;             // timing experiment: no pair loop, no write-out
  for (int b0 = 0; b0 < ns; b0 += 64) {
    const int nb = (ns - b0) < 64 ? (ns - b0) : 64;
    const int mine = (lane < nb) ? (int)(refined ? kept[b0 + lane] : surv[b0 + lane]) : 0;
    uint32_t off_cur = (uint32_t)__builtin_amdgcn_readlane(mine, 0) * (uint32_t)sizeof(SegFrame);
    u32x16 ra, rb;
    smem_load16(ra, frames, off_cur);
    smem_wait(ra);
    int k = 0;
    for (; k + 1 < nb; k += 2) {
      const uint32_t off1 = (uint32_t)__builtin_amdgcn_readlane(mine, k + 1) * (uint32_t)sizeof(SegFrame);
      const int k2 = (k + 2 < nb) ? k + 2 : nb - 1;
      const uint32_t off2 = (uint32_t)__builtin_amdgcn_readlane(mine, k2) * (uint32_t)sizeof(SegFrame);
      pair_step(frames, ra, rb, off_cur, off1, r, extinction, sum);
      pair_step(frames, rb, ra, off1, off2, r, extinction, sum);
      off_cur = off2;
    }
    if (k < nb) pair_step(frames, ra, rb, off_cur, off_cur, r, extinction, sum);
  }
  {
    // pair_step leaves the second and third channel as what they have over the first, the first in the common sum: as
    // item_writeout takes them (a term count per plane, the anchors to subtract)
    const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
    sum.nc = (sum.ai[1] | sum.ai[2]) ? 1u : 0u;
    sum.ai[0] = sum.nc ? magic_bits : 0ull; sum.ai[1] += sum.nc ? magic_bits : 0ull; sum.ai[2] += sum.nc ? magic_bits : 0ull;
  }
  item_writeout<FINALIZE>(p, tile, lane, sum);
}

// Which of an item's candidates can touch its tile is a function of the segment set and the frame geometry, like the
// lists and the queue themselves: the frame that sizes the workspace for them also runs the test once for every item
// (two passes over the queue: count, prefix sum on the host's side of the launch, write) and leaves the survivors
// behind the queue; every later frame of that segment set and geometry starts its items at the pair loop.  One wave per
// item; the rays' row of tile planes must be in place (any frame of this geometry and scene has written it).
template <bool WRITE>
__global__ __launch_bounds__(64) void refine_items_kernel(const RenderParams p) {
  __shared__ uint32_t surv[PB_SLICE];
  const int lane = threadIdx.x;
  const uint32_t n_items = (uint32_t)p.n_items_host < (uint32_t)p.item_capacity ? (uint32_t)p.n_items_host : (uint32_t)p.item_capacity;
  const uint32_t c = blockIdx.x;
  if (c >= n_items) return;
  const PB_CONST_AS uint32_t* it = pb_uniform(p.items + (size_t)c * 4);
  const int group = (int)it[0];
  const uint32_t start = it[1];
  const int cn = (int)it[2];
  const uint32_t refined = it[3];
  const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  V3 pl[4];
  {
    const PB_CONST_AS float* w = pb_uniform(rr + 7 * 64);
    pl[0] = v3(w[0], w[1], w[2]); pl[1] = v3(w[3], w[4], w[5]); pl[2] = v3(w[6], w[7], w[8]); pl[3] = v3(w[9], w[10], w[11]);
  }
  const int ns = refine_candidates<false>(p, start, cn, pl, surv, lane, refined);
  if (!WRITE) {
    if (lane == 0) p.surv_cnt[c] = (uint32_t)ns;
    return;
  }
  __syncthreads();
  const uint32_t at = pb_uniform(p.surv_cnt + n_items)[c];
  for (int k = lane; k < ns; k += 64) p.surv_list[at + k] = surv[k];
}

// The kept queue, cut anew by survivors.  A tile's items are neighbours in the queue, so their survivors are one
// contiguous run of surv_list: [off[first item], off[first item + slices]).  That run is cut into items of at most
// `length` survivors - every item of the kept queue then carries about the same number of pair-loop steps, where a slice
// of a macro tile's list left anything from none to all of its candidates.  One lane per tile.  WRITE = false counts
// the new items per queue slot (regroup_cnt, zeroed before); WRITE = true writes them at regroup_off[slot] into
// items_kept and the tile's count into tile_items.
template <bool WRITE>
__global__ __launch_bounds__(256) void regroup_items_kernel(const RenderParams p, int n_tiles, int n_old, int length,
                                                            uint32_t* __restrict__ regroup, uint32_t* __restrict__ items_kept) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const uint32_t* off = p.surv_cnt + n_old;           // n_old + 1 entries: the last one is the total
  const uint32_t slices = pb_item_slices(p, p.tile_count[t]);
  const int nsub = p.ssaa * p.ssaa;
  int m;
  const int slot = tile_slot(p, t, m);
  uint32_t c0 = p.item_off[slot];                      // the tile's first item in the queue as it was built
  uint32_t n_new = 0;
  const uint32_t at = WRITE ? regroup[slot] : 0u;
  for (int q = 0; q < nsub; ++q) {
    if (p.group_last[t * nsub + q] < 0) continue;
    if (c0 + slices > (uint32_t)n_old) break;          // a queue cut short by its capacity: never kept
    const uint32_t first = off[c0], total = off[c0 + slices] - first;
    for (uint32_t done = 0; done < total; done += (uint32_t)length, ++n_new) {
      if (WRITE) {
        uint32_t* dst = items_kept + (size_t)(at + n_new) * 4;
        dst[0] = (uint32_t)(t * nsub + q); dst[1] = first + done;
        dst[2] = (total - done < (uint32_t)length) ? total - done : (uint32_t)length; dst[3] = 1u;
      }
    }
    c0 += slices;
  }
  if (WRITE) p.tile_items[t] = n_new;
  else regroup[slot] = n_new;
}

// Integral_73 / Integral_73Het with relaxed arithmetic, in the same one-wave-per-item shape.  What the exact-order
// estimator spends its time on - the closest approach of beam axis and view ray in double, and for every valid pair a
// walk through the grid from the camera to that point - reduces to a few float operations here:
//  * the closest approach is the mid-point of the cylinder crossing's two roots (pair_het_fast_geometry);
//  * the walk's result depends on the view ray and on where it is cut only: each item tabulates, per ray, the running
//    sums of the samples at the crossings of each grid axis (view_walk_tabulate: 37 samples per ray), and a valid pair
//    then costs three table look-ups and the one sample at its own end point (view_walk_mean);
//  * transmittance along beam and view ray is one exp2 of the summed optical depths.
// The frame differs from the exact-order one by float rounding, and by a contribution gained or lost where a
// closest approach falls within rounding of a cap of its segment; pb_set_precise(ctx, 1) selects the exact-order kernels.
constexpr int HIT_QUEUE = 128;      // (ray, segment) pairs an item can have waiting in LDS
struct HetHalf { float r2, osy, aos, bos, A, B, krad, pad; };
static_assert(sizeof(HetHalf) == sizeof(HetFast), "one scalar load of eight dwords");

PB_DI void smem_load8(u32x8& dst, const HetFast* base, uint32_t byte_offset) {
  asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(dst) : "s"(base), "s"(byte_offset));
}
PB_DI void smem_wait(u32x16& a, u32x16& b, u32x8& c) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c)); }
// The same loads from a record's own 64-bit address (first or second half): segment sets whose records span more than
// the 2 GB a 32-bit byte offset is trusted with (11 M segments) cannot be indexed from the array's base.
PB_DI void smem_load16_first(u32x16& dst, const SegFrame* rec) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(dst) : "s"(rec)); }
PB_DI void smem_load16_second(u32x16& dst, const SegFrame* rec) { asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=&s"(dst) : "s"(rec)); }
PB_DI void smem_load8_at(u32x8& dst, const HetFast* rec) { asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(dst) : "s"(rec)); }

// A lane's own copy of a segment's record, fetched as twelve 16-byte loads: field by field, a wave whose lanes read 64
// different records issues some forty-five loads of 64 scattered dwords each.
PB_DI SegFrame load_frame_wide(const SegFrame* src) {
  SegFrame f;
  const uint4* s4 = (const uint4*)src;
  uint4* d4 = (uint4*)&f;
#pragma unroll
  for (int i = 0; i < PB_FRAME_USED_BYTES / 16; ++i) d4[i] = s4[i];
  return f;
}

struct HetItem {
  const SegFrame* frames;
  const HetFast* het;
  GridView grid;
  ViewWalk walk;
  const float* tab;        // this wave's walk tables in LDS
  float d1_len;            // |d1|: the reference cuts the view ray's walk at d1 * distance, d1 the vec4-normalised direction
  float extinction;        // sigma_t (the view ray's extinction when there is no grid)
  int lane;
  uint32_t* q_seg;         // pairs left to the exact-order code: segment and lane (ray) of each, in LDS
  uint32_t* q_ray;
};

// Returns the lanes whose pair the exact-order code has to decide (pair_het_fast_geometry; likewise an axis point
// within HET_FAST_EDGE of an end of the segment's stretch of the beam).
template <bool WIDE>
PB_DI unsigned long long pair_step_het(const HetItem& h, u32x16& cur, u32x16& nxt, uint32_t idx_cur, uint32_t idx_nxt, const Ray& r,
                                       PairSums& sum) {
  u32x16 second;
  u32x8 third;
  if (WIDE) {
    smem_load16_second(second, h.frames + idx_cur);
    smem_load8_at(third, h.het + idx_cur);
    smem_load16_first(nxt, h.frames + idx_nxt);
  } else {
    const uint32_t off_cur = idx_cur * (uint32_t)sizeof(SegFrame), off_nxt = idx_nxt * (uint32_t)sizeof(SegFrame);
    smem_load16(second, h.frames, off_cur + 64u);
    smem_load8(third, h.het, idx_cur * (uint32_t)sizeof(HetFast));
    smem_load16(nxt, h.frames, off_nxt);
  }
  SegHalfA a;
  __builtin_memcpy(&a, &cur, 64);
  float tc_minus = 0.0f, t_mid = 0.0f, y_star = 0.0f, y_err = 0.0f, sin2 = 1.0f;
  const int res = pair_het_fast_geometry(a.Rinv, a.ox2, a.oz2, a.c4, a.oy, a.len, r, tc_minus, t_mid, y_star, y_err, sin2);
  smem_wait(second, nxt, third);
  SegHalfB b;
  __builtin_memcpy(&b, &second, 64);
  HetHalf f;
  __builtin_memcpy(&f, &third, 32);
  bool ok = res == 2;
  bool defer = res == 3;
  if (ok && !pair_721_fast_sure(a.qflags, tc_minus, r.maxd)) ok = pair_721_fast_gate(b.lo, b.hi, b.cand, a.qflags, r);
  if (ok) {
    // distance of the axis point from orig_start, within [|start - orig_start|, |end - orig_start|] (BeamRadiance.h:248)
    const float dy = y_star - f.osy;
    const float beta = __builtin_amdgcn_sqrtf(__builtin_fmaf(dy, dy, f.r2));
    const float edge = HET_FAST_EDGE * f.bos + y_err;
    ok = (beta >= f.aos + edge) && (beta <= f.bos - edge);
    defer = !ok && (beta >= f.aos - edge) && (beta <= f.bos + edge);
    if (ok) {
      const float view_distance = t_mid * a.len;
      const float beam_extinction = __builtin_fmaf(f.B, beta - f.aos, f.A);
      float view_extinction = h.extinction;
      if (h.grid.z) {
        const float cut = view_distance * h.d1_len;
        defer = cut >= h.walk.t_unsure;       // the exact-order code evaluates this pair; the term added below is zero
        view_extinction = view_walk_mean(h.grid, h.walk, r.l, cut, h.tab, h.lane);
      }
      const float depth = __builtin_fmaf(beam_extinction * beta, a.len, view_extinction * view_distance);
      const float transmitted = __builtin_amdgcn_exp2f(__builtin_fmaf(depth, -1.44269504f, -10.0f));   // times the 2^-10 of the sums
      float value = transmitted * beam_extinction * f.krad * __builtin_amdgcn_rsqf(sin2);
      if (transmitted < 1e-6f * 0.0009765625f || defer) value = 0.0f;                                   // BeamRadiance.h:273
      const double integral = (double)__builtin_amdgcn_fmed3f(value, 0.0f, b.vcap10);
      if (a.qflags & 64u) {
        sum.ag += (unsigned long long)__double_as_longlong(__builtin_fma(b.W[0], integral, PB_FIX_MAGIC));
        sum.ng++;
      } else {
        sum.ai[0] += (unsigned long long)__double_as_longlong(__builtin_fma(b.W[0], integral, PB_FIX_MAGIC));
        sum.ai[1] += (unsigned long long)__double_as_longlong(__builtin_fma(b.W[1], integral, PB_FIX_MAGIC));
        sum.ai[2] += (unsigned long long)__double_as_longlong(__builtin_fma(b.W[2], integral, PB_FIX_MAGIC));
        sum.nc++;
      }
    }
  }
  return __ballot(defer && r.active);
}

// One wave per ray group: every lane walks its view ray through the grid once and leaves the running sums of the
// crossing samples (view_walk_tabulate) for the frame's evaluation items, of which a tile has many.
__global__ __launch_bounds__(64) void view_walk_kernel(const RenderParams p) {
  const int group = blockIdx.x, lane = threadIdx.x;
  const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  if (pb_uniform(p.group_last)[group] < 0) return;          // no ray of the group is active: no item reads its tables
  GridView grid;
  grid.z = p.grid_z; grid.side = p.grid_side; grid.fast_div = p.grid_fast_div;
  float* wt = p.walk + (size_t)group * VIEW_WALK_STRIDE * 64;
  const V3 l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
  ViewWalk w = view_walk_tabulate(grid, l, wt, lane);
  w.lx = l.x; w.ly = l.y;
  w.d1_len = length3(v3(rr[8 * 64 + lane], rr[9 * 64 + lane], rr[10 * 64 + lane]));
  view_walk_store(w, wt + VIEW_WALK_ROWS * 64, lane);
}

// LDS_TAB: the rays' walk tables are copied to LDS (10 KB a wave: eleven waves a CU); without, a valid pair's three
// look-ups go to the tables in global memory and four waves fit a SIMD (4096^2 rays, 23.7 M segments, eight bands:
// 2.42 -> 1.78 s a frame); five, with 96 registers and 55 dwords a lane in scratch, 1.72 s.
template <bool FINALIZE, bool WIDE, bool LDS_TAB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LDS_TAB ? 3 : 5, LDS_TAB ? 4 : 5))) void eval_items_het_kernel(const RenderParams p) {
  __shared__ uint32_t surv[PB_SLICE];
  __shared__ float tab[LDS_TAB ? VIEW_WALK_ROWS * 64 : 1];
  __shared__ uint32_t q_seg[HIT_QUEUE], q_ray[HIT_QUEUE];
  __shared__ unsigned long long s_acc[3][64];
  const int lane = threadIdx.x;
  // the number of items is the one the host measured when it sized the workspace for this segment set and frame
  // (the queue build leaves the same number in ctrl[0]): an argument instead of a trip to memory before the first load
  const uint32_t produced = (uint32_t)p.n_items_host;
  const uint32_t n_items = produced < (uint32_t)p.item_capacity ? produced : (uint32_t)p.item_capacity;
  uint32_t c = blockIdx.x;
  if (p.xcd_run > 0) {
    const uint32_t run_len = (uint32_t)p.xcd_run, xcd = c & 7u, j = c >> 3;
    c = ((j / run_len) * 8u + xcd) * run_len + j % run_len;
  }
  if (c >= n_items) return;
  const PB_CONST_AS uint32_t* it = pb_uniform(p.items + (size_t)c * 4);
  const int group = (int)it[0];
  const uint32_t start = it[1];
  const int cn = (int)it[2];
  const int tile = group / (p.ssaa * p.ssaa);
  const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  V3 pl[4];
  {
    const PB_CONST_AS float* w = pb_uniform(rr + 7 * 64);
    pl[0] = v3(w[0], w[1], w[2]); pl[1] = v3(w[3], w[4], w[5]); pl[2] = v3(w[6], w[7], w[8]); pl[3] = v3(w[9], w[10], w[11]);
  }
  Ray r;
  r.l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
  r.dn = v3(rr[3 * 64 + lane], rr[4 * 64 + lane], rr[5 * 64 + lane]);
  r.maxd = rr[6 * 64 + lane];
  r.d1 = v3(rr[8 * 64 + lane], rr[9 * 64 + lane], rr[10 * 64 + lane]);
  r.active = r.maxd >= 0.0f;
  int ns = refine_candidates<true>(p, start, cn, pl, surv, lane, it[3]);
  HetItem h;
  h.frames = p.frames; h.het = p.het_fast;
  h.grid.z = p.grid_z; h.grid.side = p.grid_side; h.grid.fast_div = p.grid_fast_div;
  h.tab = tab; h.lane = lane; h.extinction = p.extinction_c;
  h.q_seg = q_seg; h.q_ray = q_ray;
  h.d1_len = length3(r.d1);
  if (ns > 0 && h.grid.z) {
    // the rays' tabulated walks (view_walk_kernel): members to registers, tables to LDS
    const float* wt = p.walk + (size_t)group * VIEW_WALK_STRIDE * 64;
    h.walk = view_walk_load(wt + VIEW_WALK_ROWS * 64, lane);
    if (LDS_TAB) {
#pragma unroll 8
      for (int row = 0; row < VIEW_WALK_ROWS; ++row) tab[row * 64 + lane] = wt[row * 64 + lane];
    } else h.tab = wt;
  }
  s_acc[0][lane] = 0ull; s_acc[1][lane] = 0ull; s_acc[2][lane] = 0ull;
  __syncthreads();
  // The pairs left to the exact-order code wait in LDS until there are 64 of them (or the item ends): every lane then
  // takes one, whatever ray it belongs to, and adds its term to that ray's sums in LDS.
  int pending = 0;
  bool any_exact = false;
  auto push = [&](unsigned long long mask, uint32_t sidx) {
    if ((mask >> lane) & 1ull) {
      const int e = pending + (int)lanes_below(mask);
      q_seg[e] = sidx; q_ray[e] = (uint32_t)lane;
    }
    pending += __builtin_popcountll(mask);
  };
  // (the drain takes what it needs of the kernel's arguments from the argument block again, kernarg_again, instead of
  // from scalar registers that would stay occupied across the survivor loop)
  auto drain = [&](int take) {
    __syncthreads();
    if (lane < take) {
      const PB_CONST_AS RenderParams* q = kernarg_again<RenderParams>();
      const float* rr = q->rays + (size_t)group * PB_RAY_WORDS * 64;
      GridView grid;
      grid.z = q->grid_z; grid.side = q->grid_side; grid.fast_div = q->grid_fast_div;
      const int e = pending - take + lane;
      const uint32_t sidx = q_seg[e], rl = q_ray[e];
      Ray rq;
      rq.l = v3(rr[0 * 64 + rl], rr[1 * 64 + rl], rr[2 * 64 + rl]);
      rq.dn = v3(rr[3 * 64 + rl], rr[4 * 64 + rl], rr[5 * 64 + rl]);
      rq.maxd = rr[6 * 64 + rl];
      rq.d1 = v3(rr[8 * 64 + rl], rr[9 * 64 + rl], rr[10 * 64 + rl]);
      rq.active = true;
      const SegFrame f = load_frame_wide(q->frames + sidx);
      CylHit hit;
      if (hit_cylinder_exact(f, rq, hit) == 2) {
        const float4 A = q->capA[sidx];
        double integral = integral_73het(grid, v3(A.x, A.y, A.z), q->het[sidx], q->het_axis[sidx], hit, rq, q->extinction_c, q->scattering_c);
        const double W0 = f.W[0], W1 = f.W[1], W2 = f.W[2];
        integral = fmin(integral, PB_FIX_LIMIT / fmax(W0, fmax(W1, W2)));
        integral = integral * PB_FIX_PRESCALE;
        const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
        const unsigned long long v0 = (unsigned long long)__double_as_longlong(__builtin_fma(W0, integral, PB_FIX_MAGIC)) - magic_bits;
        const unsigned long long v1 = (unsigned long long)__double_as_longlong(__builtin_fma(W1, integral, PB_FIX_MAGIC)) - magic_bits;
        const unsigned long long v2 = (unsigned long long)__double_as_longlong(__builtin_fma(W2, integral, PB_FIX_MAGIC)) - magic_bits;
        if (v0) atomicAdd(&s_acc[0][rl], v0);
        if (v1) atomicAdd(&s_acc[1][rl], v1);
        if (v2) atomicAdd(&s_acc[2][rl], v2);
      }
    }
    pending -= take;
    any_exact = true;
    __syncthreads();
  };
  if (ns > 32) __builtin_amdgcn_s_setprio(3);
  else if (ns > 16) __builtin_amdgcn_s_setprio(2);
  else if (ns > 8) __builtin_amdgcn_s_setprio(1);
  PairSums sum;
  sum.ai[0] = sum.ai[1] = sum.ai[2] = sum.ag = 0ull; sum.nc = sum.ng = 0u;
  for (int b0 = 0; b0 < ns; b0 += 64) {
    const int nb = (ns - b0) < 64 ? (ns - b0) : 64;
    const int mine = (lane < nb) ? (int)surv[b0 + lane] : 0;
    uint32_t idx_cur = (uint32_t)__builtin_amdgcn_readlane(mine, 0);
    u32x16 ra, rb;
    if (WIDE) smem_load16_first(ra, h.frames + idx_cur);
    else smem_load16(ra, h.frames, idx_cur * (uint32_t)sizeof(SegFrame));
    smem_wait(ra);
    int k = 0;
    for (; k + 1 < nb; k += 2) {
      const uint32_t idx1 = (uint32_t)__builtin_amdgcn_readlane(mine, k + 1);
      const int k2 = (k + 2 < nb) ? k + 2 : nb - 1;
      const uint32_t idx2 = (uint32_t)__builtin_amdgcn_readlane(mine, k2);
      const unsigned long long m0 = pair_step_het<WIDE>(h, ra, rb, idx_cur, idx1, r, sum);
      if (m0) { push(m0, idx_cur); if (pending >= 64) drain(64); }
      const unsigned long long m1 = pair_step_het<WIDE>(h, rb, ra, idx1, idx2, r, sum);
      if (m1) { push(m1, idx1); if (pending >= 64) drain(64); }
      idx_cur = idx2;
    }
    if (k < nb) {
      const unsigned long long m = pair_step_het<WIDE>(h, ra, rb, idx_cur, idx_cur, r, sum);
      if (m) { push(m, idx_cur); if (pending >= 64) drain(64); }
    }
  }
  if (pending > 0) drain(pending);
  if (any_exact) {
    // the exact-order terms carry no anchor: they join the lane's sums as one more term
    const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
    const unsigned long long a0 = s_acc[0][lane], a1 = s_acc[1][lane], a2 = s_acc[2][lane];
    if (a0 | a1 | a2) {
      sum.ai[0] += a0 + magic_bits; sum.ai[1] += a1 + magic_bits; sum.ai[2] += a2 + magic_bits;
      sum.nc++;
    }
  }
  item_writeout<FINALIZE>(p, tile, lane, sum);
}

// The same estimator with its two halves run apart.  In eval_items_het_kernel every lane of the wave goes through the whole
// pair - cylinder crossing, then the axis point's bounds, the tabulated walk, the exponential, the fixed-point sums -
// although only about one (ray, survivor) pair in four has a crossing whose closest approach lies inside it: as long as
// one lane does, the wave issues it all.  Here the pair loop (wave-uniform record through the scalar cache, as before) ends
// with the crossing and the gate; the lanes that passed queue (segment, ray, t_mid, y*, margin, sin^2) in LDS, and whenever
// 64 are waiting every lane takes one - whatever ray it belongs to - fetches what it needs of the segment's records with
// vector loads, evaluates the rest and adds its term to that ray's sums in LDS (fixed point: the order does not matter).
// The rays' walk tables stay in global memory (three look-ups per queued pair, served by the caches; in LDS they were
// 10 KB a wave and held the kernel to three waves per SIMD).  With items of at most HET2_SLICE candidates and one word
// per pair of the exact-order queue a wave needs 7.5 KB of LDS and 95 VGPRs: five waves per SIMD (512^2, 236 629
// segments: 0.845 -> 0.769 ms a frame against four).
constexpr int HET2_SLICE = 256;       // candidates of an item of this kernel (het_two_pass): with the queues below, 7.5 KB of LDS a wave
constexpr int FAST_QUEUE = 192;       // up to 63 waiting + two survivors' worth of lanes between two looks at the queue
constexpr int EXACT_QUEUE = 192;      // likewise: 63 + two survivors, or 63 + what one drain of the fast queue hands over
// KEPT: every item is one of a kept queue (its survivors lie in surv_list and are read from there): without the list
// in LDS a wave needs 6.5 KB, and six waves fit a SIMD where five did.
template <bool FINALIZE, bool WIDE, bool KEPT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KEPT ? 6 : 5, KEPT ? 6 : 5))) void eval_items_het2_kernel(const RenderParams p) {
  __shared__ uint32_t surv[KEPT ? 1 : HET2_SLICE];
  __shared__ uint32_t fq_key[FAST_QUEUE];                 // segment * 64 + ray (lane) of a pair whose crossing holds its closest approach
  __shared__ float fq_val[4][FAST_QUEUE];                 // t_mid, y*, the margin on y*, sin^2(theta)
  __shared__ uint32_t q_pair[EXACT_QUEUE];                // pairs left to the exact-order code: segment * 64 + ray
  __shared__ unsigned long long s_acc[4][64];             // per ray: red, green, blue, and what all three share
  const int lane = threadIdx.x;
  // the number of items is the one the host measured when it sized the workspace for this segment set and frame
  // (the queue build leaves the same number in ctrl[0]): an argument instead of a trip to memory before the first load
  const uint32_t produced = (uint32_t)p.n_items_host;
  const uint32_t n_items = produced < (uint32_t)p.item_capacity ? produced : (uint32_t)p.item_capacity;
  uint32_t c = blockIdx.x;
  if (p.xcd_run > 0) {
    const uint32_t run_len = (uint32_t)p.xcd_run, xcd = c & 7u, j = c >> 3;
    c = ((j / run_len) * 8u + xcd) * run_len + j % run_len;
  }
  if (c >= n_items) return;
  const PB_CONST_AS uint32_t* it = pb_uniform(p.items + (size_t)c * 4);
  const int group = (int)it[0];
  const uint32_t start = it[1];
  const int cn = (int)it[2];
  const int tile = group / (p.ssaa * p.ssaa);
  const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
  V3 pl[4] = {};
  if (!KEPT) {
    const PB_CONST_AS float* w = pb_uniform(rr + 7 * 64);
    pl[0] = v3(w[0], w[1], w[2]); pl[1] = v3(w[3], w[4], w[5]); pl[2] = v3(w[6], w[7], w[8]); pl[3] = v3(w[9], w[10], w[11]);
  }
  Ray r;
  r.l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
  r.dn = v3(rr[3 * 64 + lane], rr[4 * 64 + lane], rr[5 * 64 + lane]);
  r.maxd = rr[6 * 64 + lane];
  r.d1 = r.dn;
  r.active = r.maxd >= 0.0f;
  const uint32_t* kept = p.surv_list + start;
  int ns = KEPT ? cn : refine_candidates<true>(p, start, cn, pl, surv, lane, it[3]);
  s_acc[0][lane] = 0ull; s_acc[1][lane] = 0ull; s_acc[2][lane] = 0ull; s_acc[3][lane] = 0ull;
  __syncthreads();
  const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
  int pending = 0, waiting = 0;          // entries of the exact-order queue, of the fast queue
  // The drains below take what they need of the kernel's arguments from the argument block again (kernarg_again) instead
  // of from registers that would stay occupied across the survivor loop, which is short of scalar registers as it is.
  auto push_exact = [&](unsigned long long mask, uint32_t sidx, uint32_t ray) __attribute__((always_inline)) {
    if ((mask >> lane) & 1ull) q_pair[pending + (int)lanes_below(mask)] = (sidx << 6) | ray;
    pending += __builtin_popcountll(mask);
  };
  auto drain_exact = [&](int take) __attribute__((always_inline)) {
    __syncthreads();
    const PB_CONST_AS RenderParams* q = kernarg_again<RenderParams>();
    const float* rr = q->rays + (size_t)group * PB_RAY_WORDS * 64;
    GridView grid;
    grid.z = q->grid_z; grid.side = q->grid_side; grid.fast_div = q->grid_fast_div;
    if (lane < take && !(q->debug & 16)) {
      const uint32_t pair = q_pair[pending - take + lane];
      const uint32_t sidx = pair >> 6, rl = pair & 63u;
      Ray rq;
      rq.l = v3(rr[0 * 64 + rl], rr[1 * 64 + rl], rr[2 * 64 + rl]);
      rq.dn = v3(rr[3 * 64 + rl], rr[4 * 64 + rl], rr[5 * 64 + rl]);
      rq.maxd = rr[6 * 64 + rl];
      rq.d1 = v3(rr[8 * 64 + rl], rr[9 * 64 + rl], rr[10 * 64 + rl]);
      rq.active = true;
      const SegFrame f = load_frame_wide(q->frames + sidx);
      CylHit hit;
      if (hit_cylinder_exact(f, rq, hit) == 2) {
        const float4 A = q->capA[sidx];
        double integral = integral_73het(grid, v3(A.x, A.y, A.z), q->het[sidx], q->het_axis[sidx], hit, rq, q->extinction_c, q->scattering_c);
        const double W0 = f.W[0], W1 = f.W[1], W2 = f.W[2];
        integral = fmin(integral, PB_FIX_LIMIT / fmax(W0, fmax(W1, W2)));
        integral = integral * PB_FIX_PRESCALE;
        const unsigned long long v0 = (unsigned long long)__double_as_longlong(__builtin_fma(W0, integral, PB_FIX_MAGIC)) - magic_bits;
        const unsigned long long v1 = (unsigned long long)__double_as_longlong(__builtin_fma(W1, integral, PB_FIX_MAGIC)) - magic_bits;
        const unsigned long long v2 = (unsigned long long)__double_as_longlong(__builtin_fma(W2, integral, PB_FIX_MAGIC)) - magic_bits;
        if (v0) atomicAdd(&s_acc[0][rl], v0);
        if (v1) atomicAdd(&s_acc[1][rl], v1);
        if (v2) atomicAdd(&s_acc[2][rl], v2);
      }
    }
    pending -= take;
    __syncthreads();
  };
  // the second half of the estimator for the top `take` entries of the fast queue, one per lane
  auto drain_fast = [&](int take) __attribute__((always_inline)) {
    __syncthreads();
    const PB_CONST_AS RenderParams* q = kernarg_again<RenderParams>();
    const float* rr = q->rays + (size_t)group * PB_RAY_WORDS * 64;
    const float* wt = q->walk ? q->walk + (size_t)group * VIEW_WALK_STRIDE * 64 : nullptr;   // the group's walk tables, then its rays' ViewWalk members
    GridView grid;
    grid.z = q->grid_z; grid.side = q->grid_side; grid.fast_div = q->grid_fast_div;
    bool defer = false;
    uint32_t sidx = 0, rl = 0;
    if (lane < take && !(q->debug & 8)) {
      const int e = waiting - take + lane;
      const uint32_t key = fq_key[e];
      sidx = (key & 0x7fffffffu) >> 6; rl = key & 63u;
      const float t_mid = fq_val[0][e], y_star = fq_val[1][e], y_err = fq_val[2][e], sin2 = fq_val[3][e];
      const float4* rec = (const float4*)(q->frames + sidx);
      bool passes = true;
      if (key & 0x80000000u) {
        // the segment's gate (HitBoundingBox entry against the ray's surface distance) for the ray this pair belongs to
        const float4 g0 = rec[5], g1 = rec[6], g2 = rec[7];       // W[2] lo.x lo.y | lo.z hi.xyz | cand.xyz vcap10
        const float lo[3] = {g0.z, g0.w, g1.x}, hi[3] = {g1.y, g1.z, g1.w}, cand[3] = {g2.x, g2.y, g2.z};
        Ray rq;
        rq.dn = v3(rr[3 * 64 + rl], rr[4 * 64 + rl], rr[5 * 64 + rl]);
        rq.maxd = rr[6 * 64 + rl];
        passes = pair_721_fast_gate(lo, hi, cand, __builtin_bit_cast(uint32_t, rec[3].w), rq);
      }
      const float4 a3 = rec[3];                               // oy, len, tb0, qflags
      const float4 b0 = rec[4];                               // W[0], W[1]
      const float2 b1 = ((const float2*)rec)[10];             // W[2]
      const float vcap10 = ((const float*)rec)[31];
      const float4* hrec = (const float4*)(q->het_fast + sidx);
      const float4 h0 = hrec[0], h1 = hrec[1];                // r2, osy, aos, bos | A, B, krad, -
      const float len = a3.y;
      const uint32_t qflags = __builtin_bit_cast(uint32_t, a3.w);
      // distance of the axis point from orig_start, within [|start - orig_start|, |end - orig_start|] (BeamRadiance.h:248)
      const float dy = y_star - h0.y;
      const float beta = __builtin_amdgcn_sqrtf(__builtin_fmaf(dy, dy, h0.x));
      const float edge = HET_FAST_EDGE * h0.w + y_err;
      const bool ok = passes && (beta >= h0.z + edge) && (beta <= h0.w - edge);
      defer = passes && !ok && (beta >= h0.z - edge) && (beta <= h0.w + edge);
      if (ok) {
        // the ray's walk numbers are asked for before anything here waits for the record's second half
        ViewWalk w = {};
        if (grid.z) w = view_walk_load(wt + VIEW_WALK_ROWS * 64, (int)rl);
        const float view_distance = t_mid * len;
        const float beam_extinction = __builtin_fmaf(h1.y, beta - h0.z, h1.x);
        float view_extinction = q->extinction_c;
        if (grid.z) {
          const float cut = view_distance * w.d1_len;
          defer = cut >= w.t_unsure;           // the exact-order code evaluates this pair; the term added below is zero
          view_extinction = view_walk_mean(grid, w, v3(w.lx, w.ly, 0.0f), cut, wt, (int)rl);
        }
        const float depth = __builtin_fmaf(beam_extinction * beta, len, view_extinction * view_distance);
        const float transmitted = __builtin_amdgcn_exp2f(__builtin_fmaf(depth, -1.44269504f, -10.0f));   // times the 2^-10 of the sums
        float value = transmitted * beam_extinction * h1.z * __builtin_amdgcn_rsqf(sin2);
        if (transmitted < 1e-6f * 0.0009765625f || defer) value = 0.0f;                                   // BeamRadiance.h:273
        const double integral = (double)__builtin_amdgcn_fmed3f(value, 0.0f, vcap10);
        const double W0 = __builtin_bit_cast(double, make_float2(b0.x, b0.y));
        if (qflags & 64u) {
          const unsigned long long v = (unsigned long long)__double_as_longlong(__builtin_fma(W0, integral, PB_FIX_MAGIC)) - magic_bits;
          if (v) atomicAdd(&s_acc[3][rl], v);
        } else {
          const double W1 = __builtin_bit_cast(double, make_float2(b0.z, b0.w)), W2 = __builtin_bit_cast(double, b1);
          const unsigned long long v0 = (unsigned long long)__double_as_longlong(__builtin_fma(W0, integral, PB_FIX_MAGIC)) - magic_bits;
          const unsigned long long v1 = (unsigned long long)__double_as_longlong(__builtin_fma(W1, integral, PB_FIX_MAGIC)) - magic_bits;
          const unsigned long long v2 = (unsigned long long)__double_as_longlong(__builtin_fma(W2, integral, PB_FIX_MAGIC)) - magic_bits;
          if (v0) atomicAdd(&s_acc[0][rl], v0);
          if (v1) atomicAdd(&s_acc[1][rl], v1);
          if (v2) atomicAdd(&s_acc[2][rl], v2);
        }
      }
    }
    waiting -= take;
    const unsigned long long dm = __ballot(defer);
    if (dm) push_exact(dm, sidx, rl);
    __syncthreads();
  };
  if (ns > 32) __builtin_amdgcn_s_setprio(3);
  else if (ns > 16) __builtin_amdgcn_s_setprio(2);
  else if (ns > 8) __builtin_amdgcn_s_setprio(1);
  // one survivor: the crossing for every lane's ray; `cur` holds the first half of its record, the gate box comes with the
  // second half, and the first half of the next survivor's record is requested into `nxt`
  auto step = [&](u32x16& cur, u32x16& nxt, uint32_t idx_cur, uint32_t idx_nxt) __attribute__((always_inline)) {
    if (WIDE) smem_load16_first(nxt, p.frames + idx_nxt);
    else smem_load16(nxt, p.frames, idx_nxt * (uint32_t)sizeof(SegFrame));
    SegHalfA a;
    __builtin_memcpy(&a, &cur, 64);
    float tc_minus, t_mid, y_star, y_err, sin2;
    bool ok, undecided;
    pair_het_fast_geometry_flat(a.Rinv, a.ox2, a.oz2, a.c4, a.oy, a.len, r, ok, undecided, tc_minus, t_mid, y_star, y_err, sin2);
    smem_wait(nxt);
    // a crossing that is not sure to pass the segment's gate (pair_721_fast_sure) has the gate run on it in the second
    // pass, where every lane holds such a pair: bit 31 of its key
    const bool gate_due = !pair_721_fast_sure(a.qflags, tc_minus, r.maxd);
    const unsigned long long km = __ballot(ok);
    if (km) {
      if (ok) {
        const int e = waiting + (int)lanes_below(km);
        fq_key[e] = (idx_cur << 6) | (uint32_t)lane | (gate_due ? 0x80000000u : 0u);
        fq_val[0][e] = t_mid; fq_val[1][e] = y_star; fq_val[2][e] = y_err; fq_val[3][e] = sin2;
      }
      waiting += __builtin_popcountll(km);
    }
    const unsigned long long dm = __ballot(undecided && r.active);
    if (dm) push_exact(dm, idx_cur, (uint32_t)lane);
  };
  // Survivors in blocks of 64 (their indices in a register, picked with v_readlane), two to an iteration; the queues are
  // looked at between iterations and drained from one place each (a drain is thousands of instructions long).
  {
    int b0 = 0, k = 0, nb = 0, mine = 0;
    uint32_t idx_cur = 0;
    u32x16 ra, rb;
    bool need_block = true;
    for (;;) {
      const bool flushing = need_block && b0 >= ns;
      if (!FINALIZE && flushing && waiting == 0 && pending > 0 && pending < 64 && kernarg_again<RenderParams>()->residue) {
        // fewer than a wave's worth left: they join one of the frame's residue queues (residue_kernel) if it has room
        const PB_CONST_AS RenderParams* pa = kernarg_again<RenderParams>();
        uint2* const residue = pa->residue;
        const uint32_t q = c & (uint32_t)(PB_RESIDUE_QUEUES - 1);
        const uint32_t cap = pb_residue_queue_capacity(pa->residue_capacity);
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd((uint32_t*)residue + q * 32u, (uint32_t)pending);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        const bool fits = at + (uint32_t)pending <= cap;
        __syncthreads();
        uint2* dst = residue + PB_RESIDUE_HEAD + (size_t)q * cap;
        if (lane < pending && at + (uint32_t)lane < cap)
          dst[at + lane] = fits ? make_uint2(q_pair[lane] >> 6, ((uint32_t)group << 6) | (q_pair[lane] & 63u)) : make_uint2(0xffffffffu, 0u);
        if (fits) { pending = 0; continue; }
      }
      if (pending >= 64 || (flushing && waiting == 0 && pending > 0)) { drain_exact(pending < 64 ? pending : 64); continue; }
      if (waiting >= 64 || (flushing && waiting > 0)) { drain_fast(waiting < 64 ? waiting : 64); continue; }
      if (flushing) break;
      if (need_block) {
        nb = (ns - b0) < 64 ? (ns - b0) : 64;
        mine = (lane < nb) ? (int)(KEPT ? kept[b0 + lane] : surv[b0 + lane]) : 0;
        idx_cur = (uint32_t)__builtin_amdgcn_readlane(mine, 0);
        if (WIDE) smem_load16_first(ra, p.frames + idx_cur);
        else smem_load16(ra, p.frames, idx_cur * (uint32_t)sizeof(SegFrame));
        smem_wait(ra);
        k = 0;
        need_block = false;
      }
      if (k + 1 < nb) {
        const uint32_t idx1 = (uint32_t)__builtin_amdgcn_readlane(mine, k + 1);
        const int k2 = (k + 2 < nb) ? k + 2 : nb - 1;
        const uint32_t idx2 = (uint32_t)__builtin_amdgcn_readlane(mine, k2);
        step(ra, rb, idx_cur, idx1);
        step(rb, ra, idx1, idx2);
        idx_cur = idx2;
        k += 2;
      } else {
        step(ra, rb, idx_cur, idx_cur);
        k += 1;
      }
      if (k >= nb) { b0 += 64; need_block = true; }
    }
  }
  __syncthreads();
  // the sums in LDS carry no anchor: they join the write-out as one term each
  PairSums sum;
  const unsigned long long a0 = s_acc[0][lane], a1 = s_acc[1][lane], a2 = s_acc[2][lane], ag = s_acc[3][lane];
  sum.nc = (a0 | a1 | a2) ? 1u : 0u;
  sum.ng = ag ? 1u : 0u;
  sum.ai[0] = a0 + (sum.nc ? magic_bits : 0ull); sum.ai[1] = a1 + (sum.nc ? magic_bits : 0ull); sum.ai[2] = a2 + (sum.nc ? magic_bits : 0ull);
  sum.ag = ag + (sum.ng ? magic_bits : 0ull);
  item_writeout<FINALIZE>(p, tile, lane, sum);
}

// Frames of the estimators whose integral is long: Integral_73Het / Integral_73 (closest approach in double, a
// walk through the grid along the view ray) and BeamRadiance's dispatch (Integral_721_ada's Riemann sum).  About
// one candidate-lane pair in four is a valid hit, so evaluating the integral inside the pair loop leaves most
// lanes idle.  Here the pair loop only runs HitCylinder / HitCone and queues the valid (ray, segment) pairs in
// LDS; whenever 64 are waiting every lane takes one, evaluates the integral and adds the result to its ray's
// sums in LDS.  The sums are fixed point, so the order of the additions does not matter and the frame equals
// eval_bins_kernel's bit for bit.
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void eval_bins_queue_kernel(const RenderParams p) {
  constexpr bool GENERIC = MODE == PB_MODE_GENERIC;
  constexpr int HIT_WORDS = GENERIC ? 9 : 6;       // entry and exit points; tc-, tc+, tb- for Integral_721(_ada)
  __shared__ uint32_t surv[PB_SLICE];
  __shared__ uint32_t q_seg[HIT_QUEUE], q_ray[HIT_QUEUE];
  __shared__ float q_hit[HIT_WORDS][HIT_QUEUE];
  __shared__ float s_ray[GENERIC ? 6 : 3][64];     // the rays' normalised directions (and l, for the 721 integrals)
  __shared__ unsigned long long s_acc[3][64];
  const int lane = threadIdx.x;
  const uint32_t produced = p.kept_queue ? (uint32_t)p.n_items_host : p.ctrl[0];   // a kept queue has been cut anew since it was built
  const uint32_t n_items = produced < (uint32_t)p.item_capacity ? produced : (uint32_t)p.item_capacity;
  const int nsub = p.ssaa * p.ssaa;
  const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
  GridView grid;
  grid.z = p.grid_z; grid.side = p.grid_side; grid.fast_div = p.grid_fast_div;
  for (uint32_t c = blockIdx.x; c < n_items; c += gridDim.x) {
    const PB_CONST_AS uint32_t* it = pb_uniform(p.items + (size_t)c * 4);
    const int group = (int)it[0];
    const uint32_t start = it[1];
    const int cn = (int)it[2];
    const int tile = group / nsub;
    V3 pl[4];
    tile_planes(p, tile % p.tiles_x, tile / p.tiles_x, pl);
    const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
    Ray r;
    r.l = v3(rr[0 * 64 + lane], rr[1 * 64 + lane], rr[2 * 64 + lane]);
    r.dn = v3(rr[3 * 64 + lane], rr[4 * 64 + lane], rr[5 * 64 + lane]);
    r.maxd = rr[6 * 64 + lane];
    r.d1 = r.dn;
    r.active = r.maxd >= 0.0f;
    s_ray[0][lane] = rr[8 * 64 + lane]; s_ray[1][lane] = rr[9 * 64 + lane]; s_ray[2][lane] = rr[10 * 64 + lane];
    if (GENERIC) { s_ray[3][lane] = r.l.x; s_ray[4][lane] = r.l.y; s_ray[5][lane] = r.l.z; }
    s_acc[0][lane] = 0ull; s_acc[1][lane] = 0ull; s_acc[2][lane] = 0ull;
    const int ns = refine_candidates<true>(p, start, cn, pl, surv, lane, it[3]);
    __syncthreads();
    int pending = 0;
    // the top `take` entries of the queue, one per lane
    auto drain = [&](int take) {
      __syncthreads();
      const bool mine = lane < take;
      uint32_t rl = 0;
      double W0 = 0.0, W1 = 0.0, W2 = 0.0, integral = 0.0;
      if (mine) {
        const int e = pending - take + lane;
        const uint32_t sidx = q_seg[e];
        rl = q_ray[e];
        CylHit h;
        h.entry = v3(q_hit[0][e], q_hit[1][e], q_hit[2][e]);
        h.exit = v3(q_hit[3][e], q_hit[4][e], q_hit[5][e]);
        Ray rq;
        rq.d1 = v3(s_ray[0][rl], s_ray[1][rl], s_ray[2][rl]);
        const SegFrame& f = p.frames[sidx];           // per-lane record: only the fields used below are loaded
        W0 = f.W[0]; W1 = f.W[1]; W2 = f.W[2];
        bool long_form = true;                        // Integral_73 / Integral_73Het
        if (GENERIC) {
          h.tc_minus = q_hit[6][e]; h.tc_plus = q_hit[7][e]; h.tb_minus = q_hit[8][e];
          rq.l = v3(s_ray[3][rl], s_ray[4][rl], s_ray[5][rl]);
          const bool cone = (f.qflags & 32u) != 0u;
          long_form = !cone && p.one_dimensional != 0;
          if (cone) integral = integral_721_ada(f, h, rq, p.extinction_c, p.scattering_c);
          else if (!long_form) integral = integral_721(f, h, rq, p.extinction_c);
        }
        if (long_form) {
          const HetFrame hf = p.het[sidx];
          const float4 A = p.capA[sidx];
          integral = integral_73het(grid, v3(A.x, A.y, A.z), hf, p.het_axis[sidx], h, rq, p.extinction_c, p.scattering_c);
        }
      }
      if (mine) {
        // 1/sin(theta) is unbounded for a beam seen end-on, 1/r^2 at a cone's apex: saturate far above the
        // display range
        integral = fmin(integral, PB_FIX_LIMIT / fmax(W0, fmax(W1, W2)));
        integral = integral * PB_FIX_PRESCALE;
        const unsigned long long v0 = (unsigned long long)__double_as_longlong(__builtin_fma(W0, integral, PB_FIX_MAGIC)) - magic_bits;
        const unsigned long long v1 = (unsigned long long)__double_as_longlong(__builtin_fma(W1, integral, PB_FIX_MAGIC)) - magic_bits;
        const unsigned long long v2 = (unsigned long long)__double_as_longlong(__builtin_fma(W2, integral, PB_FIX_MAGIC)) - magic_bits;
        if (v0) atomicAdd(&s_acc[0][rl], v0);
        if (v1) atomicAdd(&s_acc[1][rl], v1);
        if (v2) atomicAdd(&s_acc[2][rl], v2);
      }
      pending -= take;
      __syncthreads();
    };
    int sidx = ns > 0 ? __builtin_amdgcn_readfirstlane((int)surv[0]) : 0;
    for (int k = 0; k < ns; ++k) {
      const uint32_t next_idx = surv[(k + 1 < ns) ? k + 1 : k];
      const SegFrame f = fetch_frame<true>(p.frames, sidx);
      CylHit h;
      bool valid;
      if (GENERIC && (f.qflags & 32u)) {
        const HetFrame hf = pb_uniform(p.het)[sidx];
        ConeFrame cf;
        __builtin_memcpy(&cf, &hf, sizeof(cf));
        valid = hit_cone_exact(f, cf, r, h) == 2;
      } else valid = hit_cylinder_exact(f, r, h) == 2;
      const unsigned long long mask = __ballot(valid);
      if (valid) {
        const int e = pending + (int)lanes_below(mask);
        q_seg[e] = (uint32_t)sidx; q_ray[e] = (uint32_t)lane;
        q_hit[0][e] = h.entry.x; q_hit[1][e] = h.entry.y; q_hit[2][e] = h.entry.z;
        q_hit[3][e] = h.exit.x; q_hit[4][e] = h.exit.y; q_hit[5][e] = h.exit.z;
        if (GENERIC) { q_hit[6][e] = h.tc_minus; q_hit[7][e] = h.tc_plus; q_hit[8][e] = h.tb_minus; }
      }
      pending += __builtin_popcountll(mask);
      if (pending >= 64) drain(64);
      sidx = __builtin_amdgcn_readfirstlane((int)next_idx);
    }
    if (pending > 0) drain(pending);
    __syncthreads();
    const unsigned long long a0 = s_acc[0][lane], a1 = s_acc[1][lane], a2 = s_acc[2][lane];
    if (a0 | a1 | a2) {
      unsigned long long* dst = p.accum + (size_t)tile * PB_ACC_PLANES * 64 + lane;
      atomicAdd(&dst[0], a0);
      atomicAdd(&dst[64], a1);
      atomicAdd(&dst[128], a2);
    }
    __syncthreads();   // surv, the queue and the sums are rewritten by the next item
  }
}

// The pairs the items of a relaxed-arithmetic heterogeneous frame handed over (eval_items_het2_kernel), 64 to a wave:
// the exact-order estimate of each, added to its pixel's sums in memory (fixed point: in any order).
constexpr int RESIDUE_BLOCKS_PER_QUEUE = 32;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void residue_kernel(const RenderParams p) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x & (uint32_t)(PB_RESIDUE_QUEUES - 1), part = blockIdx.x / (uint32_t)PB_RESIDUE_QUEUES;
  const uint32_t cap = pb_residue_queue_capacity(p.residue_capacity);
  const uint32_t queued = pb_uniform((const uint32_t*)p.residue)[q * 32u];
  const uint32_t n = queued < cap ? queued : cap;
  const uint2* src = p.residue + PB_RESIDUE_HEAD + (size_t)q * cap;
  const int nsub = p.ssaa * p.ssaa;
  GridView grid;
  grid.z = p.grid_z; grid.side = p.grid_side; grid.fast_div = p.grid_fast_div;
  const unsigned long long magic_bits = (unsigned long long)__double_as_longlong(PB_FIX_MAGIC);
  // Pairs of one pixel are pooled in LDS before they reach its sums in memory: a pixel that looks along a beam has
  // dozens of them in a batch, and atomics on one address take about 12 ns each.  A lane's pool slot is a hash of its
  // pixel; where two pixels of a batch share a slot the slot's last writer pools and the other adds directly.
  __shared__ uint32_t s_key[64];
  __shared__ unsigned long long s_v[3][64];
  constexpr uint32_t NONE = 0xffffffffu;
  for (uint32_t base = part * 64u; base < n; base += RESIDUE_BLOCKS_PER_QUEUE * 64u) {
    const uint32_t e = base + (uint32_t)lane;
    uint32_t key = NONE;
    unsigned long long v0 = 0ull, v1 = 0ull, v2 = 0ull;
    if (e < n && !(p.debug & 4)) {
      const uint2 ent = src[e];
      const uint32_t sidx = ent.x;                       // NONE: a slot of a batch that did not fit (its item evaluated it)
      if (sidx != NONE) {
        key = ent.y;
        const uint32_t group = key >> 6, rl = key & 63u;
        const float* rr = p.rays + (size_t)group * PB_RAY_WORDS * 64;
        Ray rq;
        rq.l = v3(rr[0 * 64 + rl], rr[1 * 64 + rl], rr[2 * 64 + rl]);
        rq.dn = v3(rr[3 * 64 + rl], rr[4 * 64 + rl], rr[5 * 64 + rl]);
        rq.maxd = rr[6 * 64 + rl];
        rq.d1 = v3(rr[8 * 64 + rl], rr[9 * 64 + rl], rr[10 * 64 + rl]);
        rq.active = true;
        const SegFrame f = load_frame_wide(p.frames + sidx);
        CylHit hit;
        if (hit_cylinder_exact(f, rq, hit) == 2) {
          const float4 A = p.capA[sidx];
          double integral = (p.debug & 2) ? 1.0 : integral_73het(grid, v3(A.x, A.y, A.z), p.het[sidx], p.het_axis[sidx], hit, rq, p.extinction_c, p.scattering_c);
          const double W0 = f.W[0], W1 = f.W[1], W2 = f.W[2];
          integral = fmin(integral, PB_FIX_LIMIT / fmax(W0, fmax(W1, W2)));
          integral = integral * PB_FIX_PRESCALE;
          v0 = (unsigned long long)__double_as_longlong(__builtin_fma(W0, integral, PB_FIX_MAGIC)) - magic_bits;
          v1 = (unsigned long long)__double_as_longlong(__builtin_fma(W1, integral, PB_FIX_MAGIC)) - magic_bits;
          v2 = (unsigned long long)__double_as_longlong(__builtin_fma(W2, integral, PB_FIX_MAGIC)) - magic_bits;
        }
      }
    }
    if (p.debug & 1) continue;
    const uint32_t slot = (key ^ ((key >> 6) * 29u)) & 63u;
    s_v[0][lane] = 0ull; s_v[1][lane] = 0ull; s_v[2][lane] = 0ull;
    if (key != NONE) s_key[slot] = key;
    __syncthreads();
    if (key != NONE) {
      if (s_key[slot] == key) {
        if (v0) atomicAdd(&s_v[0][slot], v0);
        if (v1) atomicAdd(&s_v[1][slot], v1);
        if (v2) atomicAdd(&s_v[2][slot], v2);
      } else {
        unsigned long long* dst = p.accum + (size_t)((key >> 6) / (uint32_t)nsub) * PB_ACC_PLANES * 64 + (key & 63u);
        if (v0) atomicAdd(&dst[0], v0);
        if (v1) atomicAdd(&dst[64], v1);
        if (v2) atomicAdd(&dst[128], v2);
      }
    }
    __syncthreads();
    {
      const unsigned long long a0 = s_v[0][lane], a1 = s_v[1][lane], a2 = s_v[2][lane];
      if (a0 | a1 | a2) {                              // this batch pooled something in slot `lane`: s_key[lane] is its pixel
        const uint32_t k2 = s_key[lane];
        unsigned long long* dst = p.accum + (size_t)((k2 >> 6) / (uint32_t)nsub) * PB_ACC_PLANES * 64 + (k2 & 63u);
        if (a0) atomicAdd(&dst[0], a0);
        if (a1) atomicAdd(&dst[64], a1);
        if (a2) atomicAdd(&dst[128], a2);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void beam_axis_kernel(const float4* __restrict__ capA, const float4* __restrict__ capB,
                                                       const HetFrame* __restrict__ het, int n, BeamAxis* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 A = capA[i], B = capB[i];
  out[i] = beam_axis(v3(A.x, A.y, A.z), v3(B.x, B.y, B.z), het[i]);
}

__global__ __launch_bounds__(256) void finalize_kernel(const RenderParams p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)p.out_w * p.out_h;
  if (i >= n) return;
  const double inv = PB_FIX_SCALE / (double)p.ssaa;      // skeleton.cpp:361: current / (double) SSAA
  const int ox = (int)(i % (size_t)p.out_w), oy = (int)(i / (size_t)p.out_w);
  const size_t tile = (size_t)(oy >> 3) * p.tiles_x + (ox >> 3);
  const unsigned long long* a = p.accum + tile * PB_ACC_PLANES * 64 + ((oy & 7) * 8 + (ox & 7));
  for (int k = 0; k < 3; ++k) {
    const long long v = (long long)(a[k * 64] + a[3 * 64]);
    p.out[i * 3 + k] = (float)((double)v * inv);
  }
}

}  // namespace

int pb_fused_queue_macros(void) { return QUEUE_FUSED_MACROS; }

// Whether an uncounted frame of `mode` is evaluated by eval_items_fast_kernel (one wave per item, 32-bit record offsets).
static bool wide_records(const RenderParams& p) { return (size_t)p.n_seg * sizeof(SegFrame) >= ((size_t)1 << 31); }
// The two-pass kernel wins where valid crossings are a minority of a tile's (ray, survivor) pairs and the records it
// fetches per lane stay in the caches (236 629 segments at 512^2: 1.09 against 1.42 ms; 2.4 M: 12.4 against 13.6 ms); over
// the 23.7 M segments (4.5 GB of records) of the largest configuration, whose frame is mostly the cone, the one-pass
// kernel is faster (2.85 against 3.43 s).  PB_TUNE_HET1 / PB_TUNE_HET2 force one or the other.
static bool het_two_pass(const RenderParams& p) {
  if (pb_tune().het1 || p.n_seg >= (1 << 25)) return false;     // its queue keys: 25 bits of segment, 6 of ray, 1 flag
  if (p.slice > HET2_SLICE) return false;                                      // its survivor list in LDS
  return !wide_records(p) || pb_tune().het2;
}
bool pb_pipelined_eval(const RenderParams& p, int mode) {
  if (p.precise || p.eval_waves > 0 || pb_tune().old_eval) return false;
  if (mode == PB_MODE_721) return !wide_records(p);
  // eval_items_het_kernel addresses records beyond 2 GB by their own 64-bit addresses (its WIDE form)
  // eval_items_het_kernel: its walk tables hold the crossings of a grid of at most VIEW_WALK_MAX_SIDE points a side
  if (mode == PB_MODE_73 || mode == PB_MODE_73HET)
    return p.het_fast && (!p.grid_z || (p.walk && p.grid_side >= 2 && p.grid_side <= VIEW_WALK_MAX_SIDE));
  return false;
}

bool pb_residue_frame(const RenderParams& p, int mode) {
  return (mode == PB_MODE_73 || mode == PB_MODE_73HET) && p.residue && pb_pipelined_eval(p, mode) && het_two_pass(p) &&
         !pb_tune().no_residue;
}

int pb_launch_cull(const RenderParams& p, int mode, bool count, hipStream_t s) {
  (void)mode;
  const int groups = p.tiles_x * p.tiles_y * p.ssaa * p.ssaa;
  if (groups <= 0) return PB_OK;
  const bool lbvh = p.lbvh_n >= 2;
  if (count && lbvh) hipLaunchKernelGGL(cull_lbvh_block_kernel<true>, dim3(groups), dim3(CULL_THREADS), 0, s, p);
  else if (count) hipLaunchKernelGGL(cull_scan_kernel<true>, dim3(groups), dim3(64), 0, s, p);
  else if (lbvh) hipLaunchKernelGGL(cull_lbvh_block_kernel<false>, dim3(groups), dim3(CULL_THREADS), 0, s, p);
  else hipLaunchKernelGGL(cull_scan_kernel<false>, dim3(groups), dim3(64), 0, s, p);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

int pb_launch_eval_reduce(const RenderParams& p, int mode, bool count, hipStream_t s) {
  const int tiles = p.tiles_x * p.tiles_y;
  if (tiles <= 0) return PB_OK;
  int waves = p.eval_waves > 0 ? p.eval_waves : p.n_items_host;   // one wave per chunk
  if (waves > p.chunk_capacity) waves = p.chunk_capacity;
  if (waves < 1) waves = 1;
  if (mode == PB_MODE_GENERIC) {
    if (count) hipLaunchKernelGGL((eval_kernel<true, true, PB_MODE_GENERIC>), dim3(waves), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((eval_kernel<false, true, PB_MODE_GENERIC>), dim3(waves), dim3(64), 0, s, p);
  }
  else if (mode != PB_MODE_721) {      // Integral_73Het, or Integral_73 when no grid is passed
    if (count) hipLaunchKernelGGL((eval_kernel<true, true, PB_MODE_73HET>), dim3(waves), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((eval_kernel<false, true, PB_MODE_73HET>), dim3(waves), dim3(64), 0, s, p);
  }
  else if (count && p.precise) hipLaunchKernelGGL((eval_kernel<true, true, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  else if (count) hipLaunchKernelGGL((eval_kernel<true, false, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  else if (p.precise) hipLaunchKernelGGL((eval_kernel<false, true, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  else hipLaunchKernelGGL((eval_kernel<false, false, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  hipLaunchKernelGGL(reduce_kernel, dim3(tiles), dim3(64), 0, s, p);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

// Ray set-up and the binning count pass in one launch (frames whose queue is built by one block).
static bool shares_launches(const RenderParams& p, bool count) { return p.fused_queue && !count && p.n_seg > 0; }

int pb_launch_ray_setup(const RenderParams& p, bool count, hipStream_t s) {
  const int groups = p.tiles_x * p.tiles_y * p.ssaa * p.ssaa;
  if (groups <= 0) return PB_OK;
  if (shares_launches(p, count)) {
    const int n_macro = pb_macro_count(p);
    const int bin_blocks = pb_mbin_blocks(p.n_seg);
    const int ray_blocks = (groups + MBIN_THREADS / 64 - 1) / (MBIN_THREADS / 64);
    const int scene_in_lds = p.n_tri <= SCENE_LDS_MAX_TRI ? 1 : 0;
    const size_t lds = std::max(pb_mbin_lds_bytes(n_macro), scene_in_lds ? (size_t)128 * p.n_tri : (size_t)0);
    hipLaunchKernelGGL(setup_count_kernel, dim3(bin_blocks + ray_blocks), dim3(MBIN_THREADS), lds, s,
                       p, n_macro, bin_blocks, groups, scene_in_lds);
    PB_HIP(hipGetLastError());
    return PB_OK;
  }
  if (count) hipLaunchKernelGGL(ray_setup_kernel<true>, dim3(groups), dim3(64), 0, s, p);
  else hipLaunchKernelGGL(ray_setup_kernel<false>, dim3(groups), dim3(64), 0, s, p);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

// One binning pass on its own: lane = macro tile for frames whose queue one block builds, the walk over each
// segment's rectangle of macro tiles otherwise.
template <bool FILL>
static void launch_bin_pass(const RenderParams& p, int n_macro, hipStream_t s) {
  if (p.fused_queue) {
    hipLaunchKernelGGL(macro_bin_kernel<FILL>, dim3(pb_mbin_blocks(p.n_seg)), dim3(MBIN_THREADS), pb_mbin_lds_bytes(n_macro), s, p, n_macro);
    return;
  }
  const unsigned blocks = (unsigned)(((size_t)p.n_seg * PB_BIN_FAN + BIN_THREADS - 1) / BIN_THREADS);
  if (n_macro <= BIN_LDS_MACROS)
    hipLaunchKernelGGL((bin_kernel<FILL, true>), dim3(blocks), dim3(BIN_THREADS), sizeof(uint32_t) * 2 * n_macro, s, p, n_macro);
  else hipLaunchKernelGGL((bin_kernel<FILL, false>), dim3(blocks), dim3(BIN_THREADS), 0, s, p, n_macro);
}

// count pass, then the two prefix sums (entries per tile -> tile_off, ctrl[2] = total entries;
// items per tile -> item_off, ctrl[0] = total items)
int pb_launch_bin_count(const RenderParams& p, bool count, hipStream_t s) {
  const int tiles = p.tiles_x * p.tiles_y;
  if (tiles <= 0) return PB_OK;
  const int n_macro = pb_macro_count(p);
  if (p.n_seg > 0 && !shares_launches(p, count)) launch_bin_pass<false>(p, n_macro, s);
  if (p.fused_queue) {
    hipLaunchKernelGGL(queue_build_kernel, dim3(1), dim3(QUEUE_THREADS), 0, s, p, n_macro, p.n_seg > 0 ? pb_mbin_blocks(p.n_seg) : 0);
  } else {
    hipLaunchKernelGGL(macro_count_kernel, dim3((n_macro + 255) / 256), dim3(256), 0, s, p, n_macro);
    hipLaunchKernelGGL(macro_rank_kernel, dim3((n_macro + 255) / 256), dim3(256), 0, s, p, n_macro);
    hipLaunchKernelGGL(tile_items_kernel, dim3((tiles + 255) / 256), dim3(256), 0, s, p, tiles);
  }
  PB_HIP(hipGetLastError());
  return PB_OK;
}

// The evaluation kernel of `mode` over the item queue, then the pixel write-out.
static int launch_eval_finalize(const RenderParams& p, int mode, bool count, hipStream_t s) {
  // one wave per item when PB_TUNE_WAVES=0: the hardware dispatcher then hands out the queue in order
  int waves = p.eval_waves > 0 ? p.eval_waves : p.n_items_host;
  if (waves > p.item_capacity) waves = p.item_capacity;
  if (waves < 1) waves = 1;
  if (mode == PB_MODE_GENERIC) {
    if (count) hipLaunchKernelGGL((eval_bins_kernel<true, true, PB_MODE_GENERIC>), dim3(waves), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(eval_bins_queue_kernel<PB_MODE_GENERIC>, dim3(waves), dim3(64), 0, s, p);
  }
  else if (mode != PB_MODE_721) {      // Integral_73Het, or Integral_73 when no grid is passed
    if (count) hipLaunchKernelGGL((eval_bins_kernel<true, true, PB_MODE_73HET>), dim3(waves), dim3(64), 0, s, p);
    else if (pb_pipelined_eval(p, mode)) {
      if (p.grid_z) hipLaunchKernelGGL(view_walk_kernel, dim3(p.tiles_x * p.tiles_y * p.ssaa * p.ssaa), dim3(64), 0, s, p);
      if (p.xcd_run > 0) waves = (waves + 8 * p.xcd_run - 1) / (8 * p.xcd_run) * (8 * p.xcd_run);    // whole runs for every XCD
      const bool wide = wide_records(p) || pb_tune().wide;
      const bool two_pass = het_two_pass(p);
      const bool lds_tab = pb_tune().het1_lds_tab;       // the one-pass kernel with its walk tables in LDS, as it was (for A/B)
      if (pb_residue_frame(p, mode)) {
        // the items hand the pairs they leave to the exact-order code to one more launch; the pixels are written after it
        if (wide) hipLaunchKernelGGL((eval_items_het2_kernel<false, true>), dim3(waves), dim3(64), 0, s, p);
        else if (p.kept_queue && !pb_tune().het_lds_list) hipLaunchKernelGGL((eval_items_het2_kernel<false, false, true>), dim3(waves), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((eval_items_het2_kernel<false, false>), dim3(waves), dim3(64), 0, s, p);
        hipLaunchKernelGGL(residue_kernel, dim3(PB_RESIDUE_QUEUES * RESIDUE_BLOCKS_PER_QUEUE), dim3(64), 0, s, p);
      } else if (p.finalize_in_eval) {
        if (two_pass && wide) hipLaunchKernelGGL((eval_items_het2_kernel<true, true>), dim3(waves), dim3(64), 0, s, p);
        else if (two_pass) hipLaunchKernelGGL((eval_items_het2_kernel<true, false>), dim3(waves), dim3(64), 0, s, p);
        else if (wide && lds_tab) hipLaunchKernelGGL((eval_items_het_kernel<true, true, true>), dim3(waves), dim3(64), 0, s, p);
        else if (wide) hipLaunchKernelGGL((eval_items_het_kernel<true, true, false>), dim3(waves), dim3(64), 0, s, p);
        else if (lds_tab) hipLaunchKernelGGL((eval_items_het_kernel<true, false, true>), dim3(waves), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((eval_items_het_kernel<true, false, false>), dim3(waves), dim3(64), 0, s, p);
        PB_HIP(hipGetLastError());
        return PB_OK;
      }
      else if (two_pass && wide) hipLaunchKernelGGL((eval_items_het2_kernel<false, true>), dim3(waves), dim3(64), 0, s, p);
      else if (two_pass) hipLaunchKernelGGL((eval_items_het2_kernel<false, false>), dim3(waves), dim3(64), 0, s, p);
      else if (wide && lds_tab) hipLaunchKernelGGL((eval_items_het_kernel<false, true, true>), dim3(waves), dim3(64), 0, s, p);
      else if (wide) hipLaunchKernelGGL((eval_items_het_kernel<false, true, false>), dim3(waves), dim3(64), 0, s, p);
      else if (lds_tab) hipLaunchKernelGGL((eval_items_het_kernel<false, false, true>), dim3(waves), dim3(64), 0, s, p);
      else hipLaunchKernelGGL((eval_items_het_kernel<false, false, false>), dim3(waves), dim3(64), 0, s, p);
    }
    else hipLaunchKernelGGL(eval_bins_queue_kernel<PB_MODE_73HET>, dim3(waves), dim3(64), 0, s, p);
  }
  else if (count && p.precise) hipLaunchKernelGGL((eval_bins_kernel<true, true, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  else if (count) hipLaunchKernelGGL((eval_bins_kernel<true, false, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  else if (p.precise) hipLaunchKernelGGL((eval_bins_kernel<false, true, PB_MODE_721>), dim3(waves), dim3(64), 0, s, p);
  else if (pb_pipelined_eval(p, mode)) {
    if (p.xcd_run > 0) waves = (waves + 8 * p.xcd_run - 1) / (8 * p.xcd_run) * (8 * p.xcd_run);    // whole runs for every XCD
    if (p.inline_item_blocks > 0) {
      // the whole frame: items that set up their own rays, then the blocks that blacken the tiles without items
      RenderParams q = p;
      q.inline_item_blocks = waves;
      const int tiles = p.tiles_x * p.tiles_y;
      hipLaunchKernelGGL((eval_items_fast_kernel<true, true>), dim3(waves + (tiles + 63) / 64), dim3(64), 0, s, q);
      PB_HIP(hipGetLastError());
      return PB_OK;
    }
    if (p.finalize_in_eval) {
      hipLaunchKernelGGL(eval_items_fast_kernel<true>, dim3(waves), dim3(64), 0, s, p);
      PB_HIP(hipGetLastError());
      return PB_OK;
    }
    hipLaunchKernelGGL(eval_items_fast_kernel<false>, dim3(waves), dim3(64), 0, s, p);
  }
  else hipLaunchKernelGGL(eval_bins_fast_kernel, dim3(waves), dim3(64), 0, s, p);
  const size_t px = (size_t)p.out_w * p.out_h;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, p);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

int pb_launch_bin_fill_eval(const RenderParams& p, int mode, bool count, hipStream_t s) {
  const int tiles = p.tiles_x * p.tiles_y;
  if (tiles <= 0) return PB_OK;
  if (shares_launches(p, count)) {
    const int n_macro = pb_macro_count(p);
    const int bin_blocks = pb_mbin_blocks(p.n_seg);
    const int item_blocks = (tiles + ITEM_LANES - 1) / ITEM_LANES;
    hipLaunchKernelGGL(fill_items_kernel, dim3(bin_blocks + item_blocks), dim3(MBIN_THREADS), pb_mbin_lds_bytes(n_macro), s,
                       p, n_macro, bin_blocks, tiles);
  } else {
    if (p.n_seg > 0) launch_bin_pass<true>(p, pb_macro_count(p), s);
    hipLaunchKernelGGL(make_items_kernel, dim3((tiles + 255) / 256), dim3(256), 0, s, p, tiles);
  }
  return launch_eval_finalize(p, mode, count, s);
}

// A frame whose list offsets (sub_off), item queue (items, ctrl[0]) and per-tile list lengths are in the workspace
// from the frame that sized it for the same segment set and frame geometry: three launches.
int pb_max_item_survivors(void) {
  static_assert(HET2_SLICE <= PB_SLICE, "eval_items_het2_kernel has the shortest survivor list");
  return HET2_SLICE;
}

int pb_launch_refine_items(const RenderParams& p, bool write, hipStream_t s) {
  if (p.n_items_host <= 0) return PB_OK;
  if (write) hipLaunchKernelGGL(refine_items_kernel<true>, dim3(p.n_items_host), dim3(64), 0, s, p);
  else hipLaunchKernelGGL(refine_items_kernel<false>, dim3(p.n_items_host), dim3(64), 0, s, p);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

int pb_launch_regroup_items(const RenderParams& p, bool write, int n_old, int length, uint32_t* regroup, uint32_t* items_kept, hipStream_t s) {
  const int tiles = p.tiles_x * p.tiles_y;
  if (tiles <= 0) return PB_OK;
  if (write) hipLaunchKernelGGL(regroup_items_kernel<true>, dim3((tiles + 255) / 256), dim3(256), 0, s, p, tiles, n_old, length, regroup, items_kept);
  else hipLaunchKernelGGL(regroup_items_kernel<false>, dim3((tiles + 255) / 256), dim3(256), 0, s, p, tiles, n_old, length, regroup, items_kept);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

int pb_launch_fast_frame(const RenderParams& p, int mode, bool lists_in_place, hipStream_t s, int stage) {
  const int groups = p.tiles_x * p.tiles_y * p.ssaa * p.ssaa;
  if (groups <= 0) return PB_OK;
  if (stage == 2) return launch_eval_finalize(p, mode, false, s);
  const int n_macro = pb_macro_count(p);
  // the lists themselves are a function of the segment set and the frame geometry too: where an earlier frame of both
  // left them in the workspace, no block bins
  const int bin_blocks = lists_in_place ? 0 : pb_mbin_blocks(p.n_seg);
  const int ray_blocks = (groups + MBIN_THREADS / 64 - 1) / (MBIN_THREADS / 64);
  const int scene_in_lds = p.n_tri <= SCENE_LDS_MAX_TRI ? 1 : 0;
  const size_t lds = std::max(lists_in_place ? (size_t)0 : pb_mbin_lds_bytes(n_macro), scene_in_lds ? (size_t)128 * p.n_tri : (size_t)0);
  if (lists_in_place) {
    // ray set-up alone: blocks of eight waves (512^2, 1052 segments: 0.0392 ms a frame with one wave to a block, 0.0386
    // with two, 0.0380 with four, 0.0375 with eight, 0.0378 with sixteen; a block stages the scene in LDS once)
    int waves = 8;
    if (pb_tune().raywaves > 0) waves = pb_tune().raywaves;
    if (waves == 4) hipLaunchKernelGGL(setup_rays_kernel<256>, dim3((groups + 3) / 4), dim3(256), lds, s, p, groups, scene_in_lds);
    else if (waves == 16) hipLaunchKernelGGL(setup_rays_kernel<1024>, dim3((groups + 15) / 16), dim3(1024), lds, s, p, groups, scene_in_lds);
    else hipLaunchKernelGGL(setup_rays_kernel<512>, dim3((groups + 7) / 8), dim3(512), lds, s, p, groups, scene_in_lds);
    if (stage == 1) { PB_HIP(hipGetLastError()); return PB_OK; }
    return launch_eval_finalize(p, mode, false, s);
  }
  hipLaunchKernelGGL(setup_fill_kernel, dim3(bin_blocks + ray_blocks), dim3(MBIN_THREADS), lds, s,
                     p, n_macro, bin_blocks, groups, scene_in_lds);
  if (stage == 1) { PB_HIP(hipGetLastError()); return PB_OK; }
  return launch_eval_finalize(p, mode, false, s);
}

int pb_launch_beam_axis(const float4* d_capA, const float4* d_capB, const HetFrame* d_het, int n, BeamAxis* d_out, hipStream_t s) {
  if (n <= 0) return PB_OK;
  hipLaunchKernelGGL(beam_axis_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_capA, d_capB, d_het, n, d_out);
  PB_HIP(hipGetLastError());
  return PB_OK;
}
