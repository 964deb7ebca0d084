// How much of an XCD's L2 a set of 128-byte lines can use when the lines sit 256 bytes apart (only every other line of
// the address range is ever touched) against lines packed back to back.  Every workgroup sweeps the same N lines over
// and over, so each XCD's L2 is asked to hold all N; the rate per touched byte falls off where they stop fitting.
//   make -C tools/micro l2_stride && tools/micro/l2_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// mode 0: line i at i * 128; 1: at i * 256; 2: at i * 256 + 128 * (i & 1) (both parities of the line index in use)
__global__ __launch_bounds__(256) void sweep(const uint4* __restrict__ base, unsigned n_lines, int mode, int passes, unsigned* out) {
  const unsigned sub = threadIdx.x & 7u, grp = threadIdx.x >> 3;             // 8 lanes x 16 bytes = one line, 32 lines a block step
  unsigned i = (blockIdx.x * 977u + grp) % n_lines;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned steps = (unsigned)passes * (n_lines / 32u);
  for (unsigned s = 0; s < steps; ++s) {
    size_t byte = mode == 0 ? (size_t)i * 128u : (size_t)i * 256u + (mode == 2 ? ((i & 1u) << 7) : 0u);
    const uint4 v = base[byte / 16u + sub];
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    i += 32u; if (i >= n_lines) i -= n_lines;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
int main() {
  const size_t bytes = 512ull << 20;
  uint4* d; unsigned* out;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"packed (128 B apart)", "256 B apart", "256 B apart, odd lines shifted by 128"};
  for (int mode = 0; mode < 3; ++mode) {
    std::printf("%s\n", names[mode]);
    for (double mb : {0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 8.0, 12.0, 16.0, 32.0}) {
      const unsigned n_lines = (unsigned)(mb * 1048576.0 / 128.0) / 32u * 32u;
      const int blocks = 2048, passes = mb < 4 ? 16 : 6;
      hipLaunchKernelGGL(sweep, dim3(blocks), dim3(256), 0, 0, d, n_lines, mode, 2, out);          // warm
      hipEventRecord(e0);
      hipLaunchKernelGGL(sweep, dim3(blocks), dim3(256), 0, 0, d, n_lines, mode, passes, out);
      hipEventRecord(e1);
      if (hipEventSynchronize(e1) != hipSuccess) return 2;
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double touched = (double)blocks * passes * (n_lines / 32u) * 32.0 * 128.0;
      std::printf("  %5.1f MB of lines: %7.3f ms, %7.1f GB/s of lines read\n", mb, ms, touched / ms * 1e-6);
    }
  }
  return 0;
}
