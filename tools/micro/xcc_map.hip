// Which XCD a workgroup of a one-wave-per-block launch runs on, against its block index.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(unsigned* out, int spin) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;       // keeps the wave resident for a while
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hwid + (x < 0 ? 1u : 0u); }
}
int main() {
  for (int n : {4096, 65536}) {
    unsigned* d; hipMalloc(&d, sizeof(unsigned) * 2 * n);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
    int match = 0; int per[16] = {0};
    for (int b = 0; b < n; ++b) { match += (h[2 * b] == (unsigned)(b % 8)); per[h[2 * b] & 15]++; }
    printf("grid %d: xcc == block %% 8 for %d blocks (%.1f %%); per xcc:", n, match, 100.0 * match / n);
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("\nfirst 32 blocks:");
    for (int b = 0; b < 32; ++b) printf(" %u", h[2 * b]);
    printf("\nblocks 32768..32799:");
    if (n > 40000) for (int b = 32768; b < 32800; ++b) printf(" %u", h[2 * b]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
