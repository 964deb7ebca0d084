// Vector-ALU issue rate on this GPU: how many cycles a wave64 v_fma_f32 / v_pk_fma_f32 / v_mul_f32 / v_cndmask /
// v_fma_f64 costs a SIMD when 1, 2, 4 or 8 waves per SIMD issue independent instructions back to back.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void rate_kernel(float* out, int iters, float seed, int active) {
  if ((int)(threadIdx.x & 63) >= active) return;    // lanes past `active` leave: the wave runs on with part of EXEC
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
  const float m = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {   // 16 independent v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 1) {   // 8 independent v_pk_fma_f32 (16 fused multiply-adds per lane)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
                        : "v"(*(double*)&b0), "v"(*(double*)&b2));)
    } else if (KIND == 2) {   // v_cndmask / v_cmp style: 8 v_max_f32
      REP8(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                        "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 3) {   // v_fma_f64
      REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m), "v"((double)c));)
    } else if (KIND == 4) {   // v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 6) {   // v_fmac_f32 (VOP2), all VGPR
      REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                        "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 7) {   // v_fmac_f32 (VOP2), SGPR src0
      REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                        "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(c));)
    } else if (KIND == 8) {   // v_mul_f32 (VOP2), all VGPR
      REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                        "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 9) {   // v_mul_f32 (VOP2), SGPR src0
      REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                        "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));)
    } else if (KIND == 10) {   // v_cmp_lt_f32 + v_cndmask_b32 pairs
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                        "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
    } else if (KIND == 11) {   // v_add_f32 (VOP2), all VGPR
      REP8(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                        "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (KIND == 5) {   // v_fma_f32 with an SGPR operand
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(c));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (float)(d0 + d1 + d2 + d3);
}

template <int KIND>
void run(const char* name, int instr_per_iter, int cus, double ghz, int active = 64) {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 8 * 4 * 64 * 4);
  const int iters = 4000;
  for (int wps : {1, 2, 4}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus, threads = 64 * 4 * wps;      // one block per CU, wps waves on each of its 4 SIMDs
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f, active);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * instr_per_iter * wps;
    std::printf("%-22s %d waves/SIMD: %.3f ms, %.2f cycles per wave instruction per SIMD (at %.2f GHz)\n", name, wps, ms,
                ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
  }
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  std::printf("%s: %d CUs, clock %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
  run<0>("v_fma_f32", 64, p.multiProcessorCount, ghz);
  run<5>("v_fma_f32 (SGPR src)", 64, p.multiProcessorCount, ghz);
  run<6>("v_fmac_f32 e32", 64, p.multiProcessorCount, ghz);
  run<7>("v_fmac_f32 e32 (SGPR)", 64, p.multiProcessorCount, ghz);
  run<8>("v_mul_f32 e32", 64, p.multiProcessorCount, ghz);
  run<9>("v_mul_f32 e32 (SGPR)", 64, p.multiProcessorCount, ghz);
  run<11>("v_add_f32 e32", 64, p.multiProcessorCount, ghz);
  run<10>("v_cmp + v_cndmask", 64, p.multiProcessorCount, ghz);
  run<1>("v_pk_fma_f32", 64, p.multiProcessorCount, ghz);
  run<2>("v_max_f32", 64, p.multiProcessorCount, ghz);
  run<3>("v_fma_f64", 64, p.multiProcessorCount, ghz);
  run<4>("v_rcp_f32", 64, p.multiProcessorCount, ghz);
  // part of EXEC: does a wave64 instruction cost less when only its low 32 (16) lanes are on?
  run<6>("v_fmac_f32, 32 lanes", 64, p.multiProcessorCount, ghz, 32);
  run<6>("v_fmac_f32, 16 lanes", 64, p.multiProcessorCount, ghz, 16);
  run<7>("v_fmac (SGPR), 32 lanes", 64, p.multiProcessorCount, ghz, 32);
  run<2>("v_max_f32, 32 lanes", 64, p.multiProcessorCount, ghz, 32);
  run<10>("v_cmp+cndmask, 32 lanes", 64, p.multiProcessorCount, ghz, 32);
  run<4>("v_rcp_f32, 32 lanes", 64, p.multiProcessorCount, ghz, 32);
  return 0;
}
