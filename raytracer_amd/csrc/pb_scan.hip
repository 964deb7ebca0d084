// Device-wide exclusive prefix sum of 32-bit counts (used to turn per-photon beam
// counts and per-beam segment counts into output offsets, so that the GPU emits
// beams and segments in the reference's sequential order).
#include "pb_internal.h"
#include "pb_scan.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// exclusive scan of one value per thread across a 256-thread block; returns the
// exclusive prefix and writes the block total to *total (all threads)
__device__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_sums, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) lds_wave_sums[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, tot = 0;
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    uint32_t s = lds_wave_sums[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wave_off + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const uint32_t* __restrict__ in,
                                                                 uint32_t* __restrict__ out, int n,
                                                                 uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0u; sum += v[k]; }
  uint32_t total;
  uint32_t prefix = block_exclusive_scan(sum, wave_sums, &total);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = prefix; prefix += v[k]; }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one block: exclusive scan of the tile sums in place, grand total to *total_out
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(uint32_t* __restrict__ tile_sums, int n_tiles,
                                                                uint32_t* __restrict__ total_out) {
  __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
  uint32_t carry = 0;
  for (int base = 0; base < n_tiles; base += SCAN_THREADS) {
    const int i = base + threadIdx.x;
    const uint32_t v = (i < n_tiles) ? tile_sums[i] : 0u;
    uint32_t total;
    const uint32_t prefix = block_exclusive_scan(v, wave_sums, &total);
    if (i < n_tiles) tile_sums[i] = carry + prefix;
    carry += total;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void add_offsets_kernel(uint32_t* __restrict__ out, int n,
                                                                  const uint32_t* __restrict__ tile_sums) {
  const uint32_t off = tile_sums[blockIdx.x];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) out[base + k] += off;
}

}  // namespace

size_t pb_scan_tmp_words(int n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1; }

int pb_exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, int n, uint32_t* d_total, uint32_t* d_tmp,
                          hipStream_t s) {
  if (n <= 0) {
    PB_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), s));
    return PB_OK;
  }
  const int tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(SCAN_THREADS), 0, s, d_in, d_out, n, d_tmp);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, d_tmp, tiles, d_total);
  hipLaunchKernelGGL(add_offsets_kernel, dim3(tiles), dim3(SCAN_THREADS), 0, s, d_out, n, d_tmp);
  PB_HIP(hipGetLastError());
  return PB_OK;
}
