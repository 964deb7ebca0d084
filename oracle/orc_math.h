// TEST INFRASTRUCTURE — part of the CPU oracle, never linked into the product.
//
// Minimal float vector/matrix types that follow the operation order of the glm
// headers vendored by the reference (glm/glm/detail/*.inl), so that predicates
// evaluate bit-for-bit like the reference's when built with -ffp-contract=off:
//   dot(vec3)   = x + y + z, left to right         func_geometric.inl:65-72
//   dot(vec4)   = (x + y) + (z + w)                func_geometric.inl:74-82
//   normalize   = v * (1 / sqrt(dot(v, v)))        func_geometric.inl:153-159, func_exponential.inl:148-153
//   cross       = (x.y*y.z - y.y*x.z, ...)         func_geometric.inl:133-142
//   reflect     = I - N * dot(N, I) * 2            func_geometric.inl:175-179
//   mat3 * vec3 = m00*x + m10*y + m20*z, L to R    type_mat3x3.inl:506-513
//   mat3 * mat3                                    type_mat3x3.inl:524-558
//   inverse(mat3) = cofactors * (1/det)            type_mat3x3.inl:36-56
//   mat4 * vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3) type_mat4x4.inl:641-652
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <cmath>

namespace orc {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct dvec3 { double x, y, z; };
struct mat3 { float m[3][3]; };  // m[col][row]
struct mat4 { float m[4][4]; };  // m[col][row]

inline vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec4 V4(float x, float y, float z, float w) { return vec4{x, y, z, w}; }
inline vec3 xyz(const vec4& v) { return vec3{v.x, v.y, v.z}; }
inline vec4 V4(const vec3& v, float w) { return vec4{v.x, v.y, v.z, w}; }

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }

inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator*(float s, vec4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
inline vec4 operator/(vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }

inline float dot(vec3 a, vec3 b) { vec3 t = a * b; return t.x + t.y + t.z; }
inline float dot(vec4 a, vec4 b) { vec4 t = a * b; return (t.x + t.y) + (t.z + t.w); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline float length(vec4 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec4 normalize(vec4 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec3 cross(vec3 x, vec3 y) {
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }

inline double ddot(dvec3 a, dvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline dvec3 operator+(dvec3 a, dvec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator-(dvec3 a, dvec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator*(dvec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator*(double s, dvec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dlength(dvec3 a) { return std::sqrt(ddot(a, a)); }
inline dvec3 dnormalize(dvec3 a) { return a * (1.0 / std::sqrt(ddot(a, a))); }
inline dvec3 D3(vec3 v) { return {(double)v.x, (double)v.y, (double)v.z}; }

inline vec3 operator*(const mat3& m, vec3 v) {
  return {m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z,
          m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
          m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z};
}
inline mat3 operator*(const mat3& a, const mat3& b) {
  mat3 r;
  for (int c = 0; c < 3; ++c)
    for (int row = 0; row < 3; ++row)
      r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
  return r;
}
inline mat3 operator+(const mat3& a, const mat3& b) {
  mat3 r;
  for (int c = 0; c < 3; ++c)
    for (int row = 0; row < 3; ++row) r.m[c][row] = a.m[c][row] + b.m[c][row];
  return r;
}
inline mat3 operator*(const mat3& a, float s) {
  mat3 r;
  for (int c = 0; c < 3; ++c)
    for (int row = 0; row < 3; ++row) r.m[c][row] = a.m[c][row] * s;
  return r;
}
inline mat3 identity3() {
  mat3 r;
  for (int c = 0; c < 3; ++c)
    for (int row = 0; row < 3; ++row) r.m[c][row] = (c == row) ? 1.0f : 0.0f;
  return r;
}
inline mat3 columns(vec3 c0, vec3 c1, vec3 c2) {
  mat3 r;
  r.m[0][0] = c0.x; r.m[0][1] = c0.y; r.m[0][2] = c0.z;
  r.m[1][0] = c1.x; r.m[1][1] = c1.y; r.m[1][2] = c1.z;
  r.m[2][0] = c2.x; r.m[2][1] = c2.y; r.m[2][2] = c2.z;
  return r;
}
inline mat3 inverse(const mat3& a) {
  const float(*m)[3] = a.m;
  float ood = 1.0f / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
                      - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
                      + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
  mat3 r;
  r.m[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
  r.m[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
  r.m[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
  r.m[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
  r.m[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
  r.m[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
  r.m[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
  r.m[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
  r.m[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
  return r;
}
inline vec4 operator*(const mat4& m, vec4 v) {
  vec4 c0{m.m[0][0], m.m[0][1], m.m[0][2], m.m[0][3]};
  vec4 c1{m.m[1][0], m.m[1][1], m.m[1][2], m.m[1][3]};
  vec4 c2{m.m[2][0], m.m[2][1], m.m[2][2], m.m[2][3]};
  vec4 c3{m.m[3][0], m.m[3][1], m.m[3][2], m.m[3][3]};
  vec4 add0 = c0 * V4(v.x, v.x, v.x, v.x) + c1 * V4(v.y, v.y, v.y, v.y);
  vec4 add1 = c2 * V4(v.z, v.z, v.z, v.z) + c3 * V4(v.w, v.w, v.w, v.w);
  return add0 + add1;
}

}  // namespace orc
#endif
