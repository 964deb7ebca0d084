// BuildTree (CastPhotons.h:800-1126) replacement: a linear BVH over the segment
// capsules, built on the GPU.
//   1. bounds of the capsule midpoints (block reduce + ordered-uint atomics)
//   2. 48-bit Morton key per segment (16 bits per axis), payload = segment index
//   3. LSD radix sort, 8-bit digits, 6 passes: per-block digit histograms in LDS,
//      a device-wide exclusive scan of (digit, block) counts, and a stable
//      scatter whose in-block ranks come from wave match masks and per-wave LDS
//      counters (no atomics on the rank path, so the permutation is stable)
//      (up to 64 sort blocks every scatter block sums the block histograms itself, and
//      the histogram of a pass is counted by the launch before it - the key kernel, then
//      each scatter for the next digit and the block a key lands in: one launch per pass)
//   4. Karras (2012) hierarchy: one lane per internal node, ties between equal
//      keys broken by sorted position
//   5. node boxes from range tables: a node's box is the union of the leaf boxes
//      over its range of the sorted order, covered by tabulated unions of 64,
//      64^2, ... leaves (no bottom-up walk, no arrival counters)
//   6. packing: each internal node record carries both child boxes, leaves are
//      the sorted capsules
// All buffers live in one allocation that grows with the segment count and is kept
// between builds; a build of a set that fits allocates nothing and does not wait for
// the device.
// The reference's tree decides nothing about the result except through the gate
// boxes stored with each segment; this structure only finds candidates.
#include <algorithm>
#include <cstring>
#include <vector>

#include "pb_internal.h"
#include "pb_scan.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;                     // keys per thread of a sort block ...
constexpr int SORT_ITEMS_SMALL = 2;               // ... and for sets that would otherwise fill few blocks
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_PASSES = 6;    // 48 bits / 8

// order-preserving map float -> uint for atomicMin/atomicMax
PB_DI uint32_t float_ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
PB_HDI float ord_float(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  float f;
  memcpy(&f, &u, sizeof(f));
  return f;
}

// Start values of the bounds, and zeros in the histograms that the key kernel and the scatter passes count into.
__global__ __launch_bounds__(256) void build_init_kernel(uint32_t* __restrict__ bounds6, uint32_t* __restrict__ hist, int hist_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6) bounds6[i] = i < 3 ? 0xFFFFFFFFu : 0u;
  for (int k = i; k < hist_words; k += gridDim.x * blockDim.x) hist[k] = 0u;
}

__global__ __launch_bounds__(256) void bounds_kernel(const float4* __restrict__ capA, const float4* __restrict__ capB,
                                                    int n, uint32_t* __restrict__ bounds6) {
  float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 a = capA[i], b = capB[i];
    const float m[3] = {(a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f, (a.z + b.z) * 0.5f};
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], m[k]); hi[k] = fmaxf(hi[k], m[k]); }
  }
  for (int k = 0; k < 3; ++k)
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_down(lo[k], off, 64));
      hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off, 64));
    }
  // one set of atomics per block: they all hit the same six words
  __shared__ float red[6][4];
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; ++k) { red[k][threadIdx.x >> 6] = lo[k]; red[3 + k][threadIdx.x >> 6] = hi[k]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = red[k][0];
    for (int w = 1; w < 4; ++w) v = (k < 3) ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
    if (k < 3) atomicMin(&bounds6[k], float_ord(v));
    else atomicMax(&bounds6[k], float_ord(v));
  }
}

PB_HDI uint64_t expand16(uint32_t v) {   // bit k of v -> bit 3k
  uint64_t x = v & 0xFFFFu;
  x = (x | (x << 16)) & 0x000000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ __launch_bounds__(256) void morton_kernel(const float4* __restrict__ capA, const float4* __restrict__ capB,
                                                    int n, const uint32_t* __restrict__ bounds6,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                    uint32_t* __restrict__ hist0, int n_blocks, int tile) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 a = capA[i], b = capB[i];
  const float m[3] = {(a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f, (a.z + b.z) * 0.5f};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    const float lo = ord_float(bounds6[k]), hi = ord_float(bounds6[3 + k]);
    const float ext = hi - lo;
    const float scale = (ext > 0.0f) ? (65535.0f / ext) : 0.0f;
    float t = (m[k] - lo) * scale;
    t = fminf(fmaxf(t, 0.0f), 65535.0f);
    q[k] = (uint32_t)t;
  }
  const uint64_t key = (expand16(q[0]) << 2) | (expand16(q[1]) << 1) | expand16(q[2]);
  keys[i] = key;
  vals[i] = (uint32_t)i;
  // the first sort pass' (digit, block) counts, where the scatter blocks sum them themselves
  if (hist0) atomicAdd(&hist0[(size_t)((uint32_t)key & 255u) * n_blocks + i / tile], 1u);
}

// ---- radix sort ----
template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const uint64_t* __restrict__ keys, int n, int shift,
                                                                int n_blocks, uint32_t* __restrict__ ghist) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * SORT_THREADS * ITEMS;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = base + k * SORT_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  ghist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = hist[threadIdx.x];
}

// lanes of the wave holding the same 8-bit digit (invalid lanes match nobody)
PB_DI unsigned long long match_digit(uint32_t digit, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// SCANNED: gscan holds the device-wide exclusive scan of the (digit, block) counts.  Otherwise it holds the counts
// themselves and every block sums what lies before its own (few blocks: cheaper than three scan launches per pass).
// next_hist: where the next pass' (digit, block) counts are gathered as the keys are placed, or null.
template <bool SCANNED, int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in, int n,
                                                                   int shift, int n_blocks,
                                                                   const uint32_t* __restrict__ gscan,
                                                                   uint64_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out,
                                                                   uint32_t* __restrict__ next_hist) {
  constexpr int TILE = SORT_THREADS * ITEMS;
  __shared__ uint32_t cnt[SORT_WAVES][256];
  __shared__ uint32_t goff[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave w owns the contiguous keys [base + w*64*ITEMS, +64*ITEMS): order (wave, item, lane) = input order.
  // They are requested first, so that their latency runs beside that of the histogram reads below.
  const int wbase = blockIdx.x * TILE + wave * 64 * ITEMS;
  uint64_t key[ITEMS];
  uint32_t val[ITEMS], rank[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = wbase + k * 64 + lane;
    key[k] = (i < n) ? keys_in[i] : 0ull;
    val[k] = (i < n) ? vals_in[i] : 0u;
  }
  for (int w = 0; w < SORT_WAVES; ++w) cnt[w][threadIdx.x] = 0;
  if (SCANNED) {
    goff[threadIdx.x] = gscan[(size_t)threadIdx.x * n_blocks + blockIdx.x];
  } else {
    // keys with digit d go after all keys with smaller digits and after the keys with digit d of earlier blocks
    static_assert(SORT_THREADS == 256, "one thread per digit");
    const uint32_t* row = gscan + (size_t)threadIdx.x * n_blocks;
    uint32_t total = 0, before = 0;
    for (int b = 0; b < n_blocks; ++b) { const uint32_t c = row[b]; before += (b < (int)blockIdx.x) ? c : 0u; total += c; }
    // exclusive scan of `total` over the 256 digits: in-wave shuffles, then the four wave sums
    uint32_t incl = total;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) cnt[0][wave] = incl;       // cnt[0][0..3] borrowed; cleared again below
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += cnt[0][w];
    __syncthreads();
    if (threadIdx.x < SORT_WAVES) cnt[0][threadIdx.x] = 0;
    goff[threadIdx.x] = base + incl - total + before;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = wbase + k * 64 + lane;
    const bool valid = i < n;
    const uint32_t digit = (uint32_t)(key[k] >> shift) & 255u;
    const unsigned long long peers = match_digit(digit, valid);
    const unsigned long long below = peers & ((1ull << lane) - 1ull);
    uint32_t prior = 0;
    if (valid) prior = cnt[wave][digit];
    rank[k] = prior + (uint32_t)__builtin_popcountll(below);
    __syncthreads();   // every peer has read the running count before its leader bumps it
    if (valid && below == 0ull) cnt[wave][digit] = prior + (uint32_t)__builtin_popcountll(peers);
    __syncthreads();
  }
  // exclusive prefix of the per-wave counts over the waves of the block, per digit
  {
    uint32_t run = 0;
    for (int w = 0; w < SORT_WAVES; ++w) { const uint32_t c = cnt[w][threadIdx.x]; cnt[w][threadIdx.x] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = wbase + k * 64 + lane;
    if (i < n) {
      const uint32_t digit = (uint32_t)(key[k] >> shift) & 255u;
      const uint32_t dst = goff[digit] + cnt[wave][digit] + rank[k];
      keys_out[dst] = key[k];
      vals_out[dst] = val[k];
      if (next_hist) atomicAdd(&next_hist[(size_t)((uint32_t)(key[k] >> (shift + 8)) & 255u) * n_blocks + dst / TILE], 1u);
    }
  }
}

// ---- hierarchy ----
PB_DI int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t x = keys[i] ^ keys[j];
  if (x == 0ull) return 64 + __builtin_clz((uint32_t)(i ^ j));
  return __builtin_clzll(x);
}

// capsule bounds with the slack the tile test uses
PB_DI void capsule_box(float4 a, float4 b, float lo[3], float hi[3]) {
  const float pad = a.w * 1.001f + 1e-5f;
  lo[0] = fminf(a.x, b.x) - pad; lo[1] = fminf(a.y, b.y) - pad; lo[2] = fminf(a.z, b.z) - pad;
  hi[0] = fmaxf(a.x, b.x) + pad; hi[1] = fmaxf(a.y, b.y) + pad; hi[2] = fmaxf(a.z, b.z) + pad;
  // uploaded records may hold infinities or NaN: an infinite bound times a zero plane component is NaN in the
  // tile test and would cull the whole subtree; keep every bound finite (a NaN record cannot contribute)
  for (int k = 0; k < 3; ++k) {
    lo[k] = fmaxf(fminf(lo[k], 1e30f), -1e30f);
    hi[k] = fmaxf(fminf(hi[k], 1e30f), -1e30f);
  }
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012), alg. 4.
// Children: >= 0 internal node, < 0 leaf ~child (sorted position).
// Thread i also writes the box of leaf i (the capsule in sorted position i): one launch less.
__global__ __launch_bounds__(256) void karras_kernel(const uint64_t* __restrict__ keys, int n, int* __restrict__ left,
                                                    int* __restrict__ right, int* __restrict__ range_lo,
                                                    int* __restrict__ range_hi, const float4* __restrict__ capA,
                                                    const float4* __restrict__ capB, const uint32_t* __restrict__ perm,
                                                    float* __restrict__ leaf_box) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t sidx = perm[i];
    float lo[3], hi[3];
    capsule_box(capA[sidx], capB[sidx], lo, hi);
    for (int k = 0; k < 3; ++k) { leaf_box[(size_t)i * 6 + k] = lo[k]; leaf_box[(size_t)i * 6 + 3 + k] = hi[k]; }
  }
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int lc = (lo == gamma) ? ~gamma : gamma;
  const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  left[i] = lc; right[i] = rc;
  range_lo[i] = lo; range_hi[i] = hi;      // the node spans the leaves lo..hi of the sorted order
}

// Node boxes without a bottom-up walk.  A node's box is the union of the leaf boxes over its range of the
// sorted order, and unions over aligned blocks of 64, 64^2, ... leaves are tabulated first, so a range is
// covered by at most 126 table entries per level.  (The usual walk - each leaf climbs, the second arrival
// at a node merges its children - needs two device-wide fences per step for boxes written on other XCDs to
// be seen; it took 13.9 ms for 1.7 M segments against 0.2 ms for everything else in the build.)
constexpr int REFIT_MAX_LEVELS = 6;      // 64^5 leaves per entry at the top: beyond any segment count
struct RefitLevels {
  const float* box[REFIT_MAX_LEVELS];    // box[0] = leaf boxes, box[l] = unions of 64^l leaves
  int n_levels;
};

// dst[b] = union of src[64 b .. 64 b + 63]: one wave per destination box.
__global__ __launch_bounds__(256) void box_union64_kernel(const float* __restrict__ src, int n_src, float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if ((size_t)b * 64 >= (size_t)n_src) return;
  const int i = b * 64 + lane;
  float v[6] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  if (i < n_src) for (int k = 0; k < 6; ++k) v[k] = src[(size_t)i * 6 + k];
  for (int off = 32; off > 0; off >>= 1)
    for (int k = 0; k < 3; ++k) {
      v[k] = fminf(v[k], __shfl_down(v[k], off, 64));
      v[3 + k] = fmaxf(v[3 + k], __shfl_down(v[3 + k], off, 64));
    }
  if (lane == 0) for (int k = 0; k < 6; ++k) dst[(size_t)b * 6 + k] = v[k];
}

__global__ __launch_bounds__(256) void node_box_kernel(const RefitLevels lv, int n, const int* __restrict__ range_lo,
                                                      const int* __restrict__ range_hi, float* __restrict__ node_box) {
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n - 1) return;
  const long long last = range_hi[node];
  long long i = range_lo[node];
  float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  while (i <= last) {
    int level = 0;
    long long size = 1;
    while (level + 1 < lv.n_levels && (i % (size * 64) == 0) && (i + size * 64 - 1 <= last)) { size *= 64; level++; }
    // Entries of this level from i on: up to the last one that lies inside the range, but not past the next
    // multiple of 64 entries, where the level is chosen anew.  Eight entries are requested before the first is
    // used (an index past the run repeats its last entry, which changes no union): the loop waits for memory
    // once per eight entries, and that wait is what the kernel's time consists of.
    const long long k0 = i / size;
    const long long k_fit = (last + 1) / size - 1;
    const long long k_end = (k0 | 63) < k_fit ? (k0 | 63) : k_fit;
    const float* base = lv.box[level];
    for (long long k = k0; k <= k_end; k += 8) {
      float v[8][6];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long e = (k + u < k_end) ? k + u : k_end;
        const float* b = base + (size_t)e * 6;
#pragma unroll
        for (int c = 0; c < 6; ++c) v[u][c] = b[c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], v[u][c]); hi[c] = fmaxf(hi[c], v[u][3 + c]); }
    }
    i = (k_end + 1) * size;
  }
  for (int k = 0; k < 3; ++k) { node_box[(size_t)node * 6 + k] = lo[k]; node_box[(size_t)node * 6 + 3 + k] = hi[k]; }
}

__global__ __launch_bounds__(256) void pack_kernel(const float4* __restrict__ capA, const float4* __restrict__ capB,
                                                  const uint32_t* __restrict__ perm, int n,
                                                  const int* __restrict__ left, const int* __restrict__ right,
                                                  const float* __restrict__ leaf_box, const float* __restrict__ node_box,
                                                  LbvhNode* __restrict__ nodes, float4* __restrict__ sA,
                                                  float4* __restrict__ sB) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t s = perm[i]; sA[i] = capA[s]; sB[i] = capB[s]; }
  if (i < n - 1) {
    const int lc = left[i], rc = right[i];
    const float* a = (lc >= 0) ? (node_box + (size_t)lc * 6) : (leaf_box + (size_t)(~lc) * 6);
    const float* b = (rc >= 0) ? (node_box + (size_t)rc * 6) : (leaf_box + (size_t)(~rc) * 6);
    LbvhNode nd;
    for (int k = 0; k < 3; ++k) { nd.lo0[k] = a[k]; nd.hi0[k] = a[3 + k]; nd.lo1[k] = b[k]; nd.hi1[k] = b[3 + k]; }
    nd.child0 = lc; nd.child1 = rc; nd.pad[0] = nd.pad[1] = 0;
    nodes[i] = nd;
  }
}

constexpr int SORT_LOCAL_SUMS_MAX_BLOCKS = 64;     // up to here a scatter block sums the block histograms itself

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

// All of the build's memory is one allocation (ctx->lb_pool), laid out for a capacity of segments: the results the
// traversal and pb_download_lbvh read, then the scratch of the build.  It grows (free + allocate, which waits for the
// device) only when a segment set exceeds the capacity; otherwise a build allocates nothing and returns as soon as
// its kernels are queued on the context's stream.  ctx->ev[5] marks the end of the last build: calls that read the
// tree on another stream wait for it there.  The build itself waits, on the device, for the preparation of the segment
// set it reads and for the frames that still read the tree it replaces (pb_wait_for_readers).
int pb_lbvh_build(pb_ctx* ctx) {
  const int n = ctx->n_seg;
  hipStream_t s = ctx->stream;
  // The build reads the capsules pb_prepare_frames wrote - on the caller's stream after pb_upload_segments_d, which does
  // not wait - and overwrites the tree a frame queued earlier on another stream may still be traversing.
  if (ctx->ev_valid[1] && ctx->prepare_stream != s) PB_HIP(hipStreamWaitEvent(s, ctx->ev[3], 0));
  {
    const int rc = pb_wait_for_readers(ctx, s, false);
    if (rc != PB_OK) return rc;
  }
  PB_HIP(hipEventRecord(ctx->ev[4], s));
  if (n < 2) {   // nothing to build: the renderer scans the (at most one) segment
    ctx->lb_n = n; ctx->lb_version = ctx->segs_version;
    PB_HIP(hipEventRecord(ctx->ev[5], s));
    ctx->ev_valid[2] = true;
    return PB_OK;
  }
  if ((size_t)n > ctx->lb_cap) {
    if (ctx->lb_pool) PB_HIP(hipFree(ctx->lb_pool));
    ctx->lb_pool = nullptr; ctx->lb_cap = 0; ctx->lb_n = -1;
    const size_t C = (size_t)n + (size_t)n / 8 + 64;
    const size_t cap_blocks = (C + SORT_TILE - 1) / SORT_TILE;
    const size_t hist_words = 256 * cap_blocks;
    const size_t sizes[] = {
        sizeof(uint64_t) * C,            // keys
        sizeof(uint32_t) * C,            // perm
        sizeof(LbvhNode) * C,            // nodes
        sizeof(float4) * C, sizeof(float4) * C,        // sA, sB
        sizeof(int) * C, sizeof(int) * C,              // left, right
        sizeof(float) * 6 * C, sizeof(float) * 6 * C,  // node_box, leaf_box
        sizeof(uint64_t) * C, sizeof(uint32_t) * C,    // scratch: keys2, vals2
        sizeof(uint32_t) * std::max(hist_words, (size_t)SORT_PASSES * 256 * SORT_LOCAL_SUMS_MAX_BLOCKS),   // (digit, block) histograms
        sizeof(uint32_t) * (pb_scan_tmp_words((int)hist_words) + 8),   // scan scratch, total
        sizeof(int) * C, sizeof(int) * C,              // range_lo, range_hi
        sizeof(float) * C,                             // unions of 64, 64^2, ... leaf boxes: 6 (C/64 + C/4096 + ...) < 0.1 C floats
        256};                                          // bounds of the midpoints
    size_t total = 0;
    for (size_t b : sizes) total += align256(b);
    PB_HIP(hipMalloc(&ctx->lb_pool, total));
    unsigned char* at = (unsigned char*)ctx->lb_pool;
    auto take = [&](size_t bytes) { unsigned char* r = at; at += align256(bytes); return (void*)r; };
    ctx->lb_keys = (uint64_t*)take(sizes[0]); ctx->lb_perm = (uint32_t*)take(sizes[1]); ctx->lb_nodes = (LbvhNode*)take(sizes[2]);
    ctx->lb_sA = (float4*)take(sizes[3]); ctx->lb_sB = (float4*)take(sizes[4]);
    ctx->lb_left = (int*)take(sizes[5]); ctx->lb_right = (int*)take(sizes[6]);
    ctx->lb_node_box = (float*)take(sizes[7]); ctx->lb_leaf_box = (float*)take(sizes[8]);
    ctx->lb_keys2 = (uint64_t*)take(sizes[9]); ctx->lb_vals2 = (uint32_t*)take(sizes[10]);
    ctx->lb_hist = (uint32_t*)take(sizes[11]); ctx->lb_scan_tmp = (uint32_t*)take(sizes[12]);
    ctx->lb_range_lo = (int*)take(sizes[13]); ctx->lb_range_hi = (int*)take(sizes[14]);
    ctx->lb_unions = (float*)take(sizes[15]); ctx->lb_bounds = (uint32_t*)take(sizes[16]);
    ctx->lb_cap = C;
  }
  // Sets of up to 64 sort blocks (of 512 keys, else of 2048) need no separate histogram and scan launches.
  const bool small = n <= SORT_LOCAL_SUMS_MAX_BLOCKS * SORT_THREADS * SORT_ITEMS_SMALL;
  const int tile = small ? SORT_THREADS * SORT_ITEMS_SMALL : SORT_TILE;
  const int n_blocks = (n + tile - 1) / tile;
  const bool merged = n_blocks <= SORT_LOCAL_SUMS_MAX_BLOCKS;
  const size_t hist_words = (size_t)256 * n_blocks;
  uint64_t* keys2 = ctx->lb_keys2;
  uint32_t* vals2 = ctx->lb_vals2;
  uint32_t* hist = ctx->lb_hist;
  uint32_t* scan_tmp = ctx->lb_scan_tmp;
  uint32_t* total = scan_tmp + pb_scan_tmp_words((int)hist_words);
  int* range_lo = ctx->lb_range_lo;
  int* range_hi = ctx->lb_range_hi;
  float* unions = ctx->lb_unions;
  uint32_t* bounds = ctx->lb_bounds;

  {
    const int zero_words = merged ? (int)(SORT_PASSES * hist_words) : 0;
    hipLaunchKernelGGL(build_init_kernel, dim3(zero_words > 0 ? (zero_words + 1023) / 1024 : 1), dim3(256), 0, s, bounds, hist, zero_words);
  }
  int bblocks = (n + 1023) / 1024;      // few blocks: each ends in atomics on the same six words
  if (bblocks > 256) bblocks = 256;
  hipLaunchKernelGGL(bounds_kernel, dim3(bblocks), dim3(256), 0, s, ctx->d_capA, ctx->d_capB, n, bounds);
  hipLaunchKernelGGL(morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->d_capA, ctx->d_capB, n, bounds,
                     ctx->lb_keys, ctx->lb_perm, merged ? hist : nullptr, n_blocks, tile);
  PB_HIP(hipGetLastError());
  uint64_t* kin = ctx->lb_keys; uint32_t* vin = ctx->lb_perm;
  uint64_t* kout = keys2; uint32_t* vout = vals2;
  for (int pass = 0; pass < SORT_PASSES; ++pass) {
    const int shift = 8 * pass;
    if (merged) {
      uint32_t* mine = hist + (size_t)pass * hist_words;
      uint32_t* next = pass + 1 < SORT_PASSES ? mine + hist_words : nullptr;
      if (small)
        hipLaunchKernelGGL((sort_scatter_kernel<false, SORT_ITEMS_SMALL>), dim3(n_blocks), dim3(SORT_THREADS), 0, s, kin, vin, n, shift,
                           n_blocks, mine, kout, vout, next);
      else
        hipLaunchKernelGGL((sort_scatter_kernel<false, SORT_ITEMS>), dim3(n_blocks), dim3(SORT_THREADS), 0, s, kin, vin, n, shift,
                           n_blocks, mine, kout, vout, next);
    } else {
      hipLaunchKernelGGL(sort_hist_kernel<SORT_ITEMS>, dim3(n_blocks), dim3(SORT_THREADS), 0, s, kin, n, shift, n_blocks, hist);
      int rc = pb_exclusive_scan_u32(hist, hist, (int)hist_words, total, scan_tmp, s);
      if (rc != PB_OK) return rc;
      hipLaunchKernelGGL((sort_scatter_kernel<true, SORT_ITEMS>), dim3(n_blocks), dim3(SORT_THREADS), 0, s, kin, vin, n, shift, n_blocks,
                         hist, kout, vout, (uint32_t*)nullptr);
    }
    PB_HIP(hipGetLastError());
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  // SORT_PASSES is even: the sorted data is back in lb_keys / lb_perm
  static_assert(SORT_PASSES % 2 == 0, "sorted keys must end in the persistent buffers");
  hipLaunchKernelGGL(karras_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->lb_keys, n, ctx->lb_left,
                     ctx->lb_right, range_lo, range_hi, ctx->d_capA, ctx->d_capB, ctx->lb_perm, ctx->lb_leaf_box);
  RefitLevels lv;
  lv.box[0] = ctx->lb_leaf_box; lv.n_levels = 1;
  {
    const float* src = ctx->lb_leaf_box;
    float* dst = unions;
    int n_src = n;
    while (n_src > 64 && lv.n_levels < REFIT_MAX_LEVELS) {
      const int n_dst = (n_src + 63) / 64;
      hipLaunchKernelGGL(box_union64_kernel, dim3((n_dst + 3) / 4), dim3(256), 0, s, src, n_src, dst);
      lv.box[lv.n_levels++] = dst;
      src = dst; dst += (size_t)n_dst * 6; n_src = n_dst;
    }
  }
  hipLaunchKernelGGL(node_box_kernel, dim3((n - 1 + 255) / 256), dim3(256), 0, s, lv, n, range_lo, range_hi, ctx->lb_node_box);
  hipLaunchKernelGGL(pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->d_capA, ctx->d_capB, ctx->lb_perm, n,
                     ctx->lb_left, ctx->lb_right, ctx->lb_leaf_box, ctx->lb_node_box, ctx->lb_nodes, ctx->lb_sA,
                     ctx->lb_sB);
  PB_HIP(hipGetLastError());
  PB_HIP(hipEventRecord(ctx->ev[5], s));
  ctx->ev_valid[2] = true;
  ctx->lb_n = n;
  ctx->lb_version = ctx->segs_version;
  return PB_OK;
}

int pb_lbvh_download(pb_ctx* ctx, uint64_t* keys, uint32_t* perm, pb_node* nodes) {
  const int n = ctx->lb_n;
  if (n < 2) return PB_OK;
  PB_HIP(hipDeviceSynchronize());
  if (keys) PB_HIP(hipMemcpy(keys, ctx->lb_keys, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
  if (perm) PB_HIP(hipMemcpy(perm, ctx->lb_perm, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
  if (nodes) {
    std::vector<int> l((size_t)n - 1), r((size_t)n - 1);
    std::vector<float> box(6 * ((size_t)n - 1));
    PB_HIP(hipMemcpy(l.data(), ctx->lb_left, sizeof(int) * l.size(), hipMemcpyDeviceToHost));
    PB_HIP(hipMemcpy(r.data(), ctx->lb_right, sizeof(int) * r.size(), hipMemcpyDeviceToHost));
    PB_HIP(hipMemcpy(box.data(), ctx->lb_node_box, sizeof(float) * box.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i + 1 < (size_t)n; ++i) {
      for (int k = 0; k < 3; ++k) { nodes[i].lo[k] = box[i * 6 + k]; nodes[i].hi[k] = box[i * 6 + 3 + k]; }
      nodes[i].left = l[i]; nodes[i].right = r[i];
    }
  }
  return PB_OK;
}
