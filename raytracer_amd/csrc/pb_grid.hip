// Batched Extinction3D (Heterogeneous.h:29-135): one lane per ray, every (distance, extinction) sample
// optionally written out.  The photon pass and the heterogeneous estimator inline the same walk
// (pb_grid.h); this entry point exposes it on its own.
#include <vector>

#include "pb_internal.h"
#include "pb_grid.h"

namespace {

__global__ __launch_bounds__(64) void extinction3d_kernel(GridView grid, const float* __restrict__ start_xyz,
                                                          const float* __restrict__ end_xyz, int n,
                                                          float* __restrict__ mean_out, int* __restrict__ count_out,
                                                          float* __restrict__ samples_out, int max_samples) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 a = v3(start_xyz[3 * i], start_xyz[3 * i + 1], start_xyz[3 * i + 2]);
  const V3 b = v3(end_xyz[3 * i], end_xyz[3 * i + 1], end_xyz[3 * i + 2]);
  DdaWalk w;
  w.begin(grid, a, b);
  float sum = 0.0f, d, e;
  int k = 0;
  while (w.next(grid, d, e)) {
    sum += e;
    if (samples_out && k < max_samples) {
      samples_out[((size_t)i * max_samples + k) * 2 + 0] = d;
      samples_out[((size_t)i * max_samples + k) * 2 + 1] = e;
    }
    k++;
  }
  mean_out[i] = sum / (float)k;
  if (count_out) count_out[i] = k;
}

struct DeviceBuf {
  void* p = nullptr;
  ~DeviceBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" int pb_extinction3d(pb_ctx* ctx, const float* start_xyz, const float* end_xyz, int n, float* mean_out,
                               int* count_out, float* samples_out, int max_samples) {
  if (!ctx || !start_xyz || !end_xyz || !mean_out || n < 0 || (samples_out && max_samples <= 0)) {
    pb_set_error("pb_extinction3d: bad argument"); return PB_ERR_ARG;
  }
  if (!ctx->d_grid) { pb_set_error("pb_extinction3d: pb_set_grid has not been called"); return PB_ERR_STATE; }
  if (n == 0) return PB_OK;
  PB_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nn = (size_t)n;
  const size_t n_samp = samples_out ? nn * (size_t)max_samples * 2 : 0;
  DeviceBuf in, out;
  PB_HIP(hipMalloc(&in.p, sizeof(float) * 6 * nn));
  PB_HIP(hipMalloc(&out.p, sizeof(float) * (2 * nn + n_samp + 2)));
  float* d_a = (float*)in.p; float* d_b = d_a + 3 * nn;
  float* d_mean = (float*)out.p; int* d_cnt = (int*)(d_mean + nn); float* d_samp = n_samp ? d_mean + 2 * nn : nullptr;
  PB_HIP(hipMemcpyAsync(d_a, start_xyz, sizeof(float) * 3 * nn, hipMemcpyHostToDevice, s));
  PB_HIP(hipMemcpyAsync(d_b, end_xyz, sizeof(float) * 3 * nn, hipMemcpyHostToDevice, s));
  if (n_samp) PB_HIP(hipMemsetAsync(d_samp, 0, sizeof(float) * n_samp, s));
  GridView g;
  g.z = ctx->d_grid; g.side = ctx->grid_side; g.fast_div = ctx->grid_fast_div;
  hipLaunchKernelGGL(extinction3d_kernel, dim3((n + 63) / 64), dim3(64), 0, s, g, d_a, d_b, n, d_mean, d_cnt, d_samp,
                     max_samples);
  PB_HIP(hipGetLastError());
  PB_HIP(hipMemcpyAsync(mean_out, d_mean, sizeof(float) * nn, hipMemcpyDeviceToHost, s));
  if (count_out) PB_HIP(hipMemcpyAsync(count_out, d_cnt, sizeof(int) * nn, hipMemcpyDeviceToHost, s));
  if (n_samp) PB_HIP(hipMemcpyAsync(samples_out, d_samp, sizeof(float) * n_samp, hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  return PB_OK;
}
