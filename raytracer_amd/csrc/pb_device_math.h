// Device-side float vector helpers whose operation order follows the glm
// headers the reference vendors (glm/glm/detail/func_geometric.inl:65-82,
// :153-159; type_mat3x3.inl:36-56, :506-558; type_mat4x4.inl:641-652).
// Every translation unit that includes this is compiled with
// -ffp-contract=off, so a*b+c below is a rounded product followed by a rounded
// sum exactly as on the reference's x86-64 build; fused forms are written
// explicitly (fmaf) only where a result is continuous in its inputs.
#ifndef PB_DEVICE_MATH_H
#define PB_DEVICE_MATH_H
#include <hip/hip_runtime.h>

#define PB_DI __device__ __forceinline__
// Read-only data that no kernel of the same launch writes can be addressed through the constant address
// space: a load whose address is wave-uniform then goes through the scalar unit (s_load) into SGPRs
// instead of sixty-four identical vector loads.
#if defined(__HIP_DEVICE_COMPILE__)
#define PB_CONST_AS __attribute__((address_space(4)))
#else
#define PB_CONST_AS   /* host pass: kernels are only parsed */
#endif
template <class T>
__device__ __forceinline__ const PB_CONST_AS T* pb_uniform(const T* p) {
  return (const PB_CONST_AS T*)(p);
}
// The argument block of a kernel whose only parameter is a T by value, as a pointer the compiler has to load through
// again: code that runs rarely inside a hot loop reads its arguments this way (a scalar load when it runs) rather than
// from scalar registers held, or spilled into vector lanes and restored, across every turn of the loop.
template <class T>
__device__ __forceinline__ const PB_CONST_AS T* kernarg_again() {
#if defined(__HIP_DEVICE_COMPILE__)
  const PB_CONST_AS T* q = (const PB_CONST_AS T*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
#else
  return nullptr;
#endif
}
#define PB_HDI __host__ __device__ __forceinline__

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M3 { float m[3][3]; };   // m[col][row]
struct D3 { double x, y, z; };

PB_HDI V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PB_HDI V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
PB_HDI V3 xyz(V4 a) { return v3(a.x, a.y, a.z); }
PB_HDI V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PB_HDI V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PB_HDI V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PB_HDI V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PB_HDI V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PB_HDI V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PB_HDI V4 operator+(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
PB_HDI V4 operator-(V4 a, V4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
PB_HDI V4 operator*(V4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
PB_HDI V4 operator*(float s, V4 a) { return v4(s * a.x, s * a.y, s * a.z, s * a.w); }
PB_HDI V4 operator/(V4 a, float s) { return v4(a.x / s, a.y / s, a.z / s, a.w / s); }

PB_HDI float pb_sqrt(float x) { return __builtin_sqrtf(x); }   // correctly rounded (hipcc default)
PB_HDI float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PB_HDI float dot4(V4 a, V4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
PB_HDI float length3(V3 a) { return pb_sqrt(dot3(a, a)); }
PB_HDI float length4(V4 a) { return pb_sqrt(dot4(a, a)); }
PB_HDI V3 normalize3(V3 a) { return a * (1.0f / pb_sqrt(dot3(a, a))); }
PB_HDI V4 normalize4(V4 a) { return a * (1.0f / pb_sqrt(dot4(a, a))); }
PB_HDI V3 cross3(V3 x, V3 y) {
  return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
PB_HDI V3 reflect3(V3 I, V3 N) { return I - N * dot3(N, I) * 2.0f; }

PB_HDI V3 mul(const M3& m, V3 v) {
  return v3(m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z,
            m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
            m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z);
}
PB_HDI M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int c = 0; c < 3; ++c)
    for (int row = 0; row < 3; ++row)
      r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
  return r;
}
PB_HDI M3 inverse3(const M3& a) {
  float ood = 1.0f / (+a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[2][1] * a.m[1][2])
                      - a.m[1][0] * (a.m[0][1] * a.m[2][2] - a.m[2][1] * a.m[0][2])
                      + a.m[2][0] * (a.m[0][1] * a.m[1][2] - a.m[1][1] * a.m[0][2]));
  M3 r;
  r.m[0][0] = +(a.m[1][1] * a.m[2][2] - a.m[2][1] * a.m[1][2]) * ood;
  r.m[1][0] = -(a.m[1][0] * a.m[2][2] - a.m[2][0] * a.m[1][2]) * ood;
  r.m[2][0] = +(a.m[1][0] * a.m[2][1] - a.m[2][0] * a.m[1][1]) * ood;
  r.m[0][1] = -(a.m[0][1] * a.m[2][2] - a.m[2][1] * a.m[0][2]) * ood;
  r.m[1][1] = +(a.m[0][0] * a.m[2][2] - a.m[2][0] * a.m[0][2]) * ood;
  r.m[2][1] = -(a.m[0][0] * a.m[2][1] - a.m[2][0] * a.m[0][1]) * ood;
  r.m[0][2] = +(a.m[0][1] * a.m[1][2] - a.m[1][1] * a.m[0][2]) * ood;
  r.m[1][2] = -(a.m[0][0] * a.m[1][2] - a.m[1][0] * a.m[0][2]) * ood;
  r.m[2][2] = +(a.m[0][0] * a.m[1][1] - a.m[1][0] * a.m[0][1]) * ood;
  return r;
}
// mat4 * vec4 with the matrix as 16 floats, column-major.
PB_HDI V4 mul4(const float* m, V4 v) {
  V4 r;
  r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * v.w);
  r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * v.w);
  r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * v.w);
  r.w = (m[3] * v.x + m[7] * v.y) + (m[11] * v.z + m[15] * v.w);
  return r;
}

#endif
