// Lane layout and arithmetic of v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4x4 outer products a wave), and whether a
// chain of three of them is, bit for bit, the v_mul_f32 / v_fmac_f32 / v_fmac_f32 chain of a 3-term dot product.
//   make -C tools/micro mfma_4x4 && tools/micro/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void layout(float* out) {
  const int l = threadIdx.x;
  // a = 100 + lane, b = 1000 * (lane + 1): D[v] of lane l = a(of some lane) * b(of some lane): decode which
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1 << (l % 4)) * (float)(1 + l / 4) * 0.0f + (float)(l + 1) * 64.0f, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
// x[k] per lane (the lane's own vector), r[i][k] per row i: chain of three MFMAs against the plain chain
__global__ __launch_bounds__(64) void chain(const float* __restrict__ rows, const float* __restrict__ x, float* out_m, float* out_v) {
  const int l = threadIdx.x;
  const int i = l & 3;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(rows[i * 3 + 0], x[l * 3 + 0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(rows[i * 3 + 1], x[l * 3 + 1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(rows[i * 3 + 2], x[l * 3 + 2], c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) {
    out_m[l * 4 + v] = c[v];
    float d = rows[v * 3 + 0] * x[l * 3 + 0];
    d = __builtin_fmaf(rows[v * 3 + 1], x[l * 3 + 1], d);
    d = __builtin_fmaf(rows[v * 3 + 2], x[l * 3 + 2], d);
    out_v[l * 4 + v] = d;
  }
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  // out = a[la] * b[lb] with a = la + 1, b = 64 (lb + 1): product / 64 = (la + 1)(lb + 1)
  int ok = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const float expect = (float)((l / 4) * 4 + v + 1) * 64.0f * (float)(l + 1);       // A from lane 4 * block + v, B from the lane itself
    ok += (h[l * 4 + v] == expect);
  }
  std::printf("D[v] of lane l == A(lane 4 * (l / 4) + v) * B(lane l): %d of 256\n", ok);
  std::printf("lane 5: %g %g %g %g (expect A of lanes 4..7 = 5..8 times B = 64 * 6)\n", h[20], h[21], h[22], h[23]);
  // chain test on awkward values (denormal products, cancellations, large magnitudes)
  std::vector<float> rows(12), x(192);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); };
  float *dr, *dx, *dm, *dv;
  hipMalloc(&dr, 48); hipMalloc(&dx, 768); hipMalloc(&dm, 1024); hipMalloc(&dv, 1024);
  long same = 0, total = 0, same_abs = 0;
  for (int trial = 0; trial < 2000; ++trial) {
    const float scale = trial % 5 == 0 ? 1e-20f : trial % 5 == 1 ? 1e18f : 1.0f;
    for (auto& r : rows) r = rnd() * (trial % 7 == 0 ? 1e-19f : 1.0f);
    for (auto& v : x) v = rnd() * scale;
    hipMemcpy(dr, rows.data(), 48, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), 768, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dr, dx, dm, dv);
    std::vector<float> m(256), v(256);
    hipMemcpy(m.data(), dm, 1024, hipMemcpyDeviceToHost); hipMemcpy(v.data(), dv, 1024, hipMemcpyDeviceToHost);
    for (int k = 0; k < 256; ++k) { total++; same += (std::memcmp(&m[k], &v[k], 4) == 0); same_abs += (m[k] == v[k]); }
  }
  std::printf("three-term chains: %ld of %ld bit-identical to mul/fma/fma (%ld equal as numbers)\n", same, total, same_abs);
  return 0;
}
