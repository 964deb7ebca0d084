// TEST INFRASTRUCTURE — CPU restatement of the reference's photon-beam path
// (see pb_oracle.h for who may use it and how it is pinned).  Every function
// cites the reference file:line it follows.  Build with -ffp-contract=off.
#include "pb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "orc_math.h"

namespace orc {

// Tessendorf.h:4 — the reference's PI is this literal, not M_PI.
static const double PI_REF = 3.14159265;
// Tessendorf.h:23
static const float ACTUAL_WIDTH = 2;
// Intersections.h:57
static const float M_BIG = FLT_MAX;

// ---------------------------------------------------------------- structs --
struct Triangle { vec4 v0, v1, v2, normal; };                       // TestModelH.h:12-39
struct Intersection { vec4 position; float distance; int index; };   // Intersections.h:7-13
struct CylIntersection {                                             // Intersections.h:15-26
  bool valid; float tb_minus, tb_plus, tc_minus, tc_plus; vec3 entry_point, exit_point;
};
struct AABB { vec4 max, min, mid; };                                 // Intersections.h:28-33
struct PhotonSeg {                                                   // Intersections.h:36-52
  vec4 start, end, mid, min, max, orig_start;
  float radius; bool ada_width; int id; float s_ext, e_ext, c_ext; int seg_id; bool scattered;
};
struct PhotonBeam {                                                  // CastPhotons.h:14-27
  vec4 start, end, omega_u, omega_v; bool ada_width; float radius; dvec3 energy;
  int index_hit; bool absorbed, scattered;
  uint32_t photon, order; bool in_scatt;  // bookkeeping only
};
struct Node {                                                        // CastPhotons.h:49-56
  AABB aabb; int seg[2]; bool leaf; int left, right; int depth;
};
struct Grid { int side_points = 0; float height = 0; std::vector<vec4> geometric_points; };  // Tessendorf.h:16-21

struct Control {
  pb_control c;
  float extinction_c() const { return c.absorption_c + c.scattering_c; }  // Control.h:48
  float focal() const { return (float)(c.screen_width * c.ssaa); }        // rasteriser.h:17
};

// ------------------------------------------------------------ scene setup --
// TestModelH.h:45-214 — Cornell box: room + short block + tall block.
static void LoadTestModel(std::vector<Triangle>& triangles) {
  triangles.clear();
  const float L = 555;
  auto P = [](float x, float y, float z) { return V4(x, y, z, 1); };
  std::vector<vec4> v0s, v1s, v2s;
  auto push = [&](vec4 a, vec4 b, vec4 c) { v0s.push_back(a); v1s.push_back(b); v2s.push_back(c); };
  {
    vec4 A = P(L, 0, 0), B = P(0, 0, 0), C = P(L, 0, L), D = P(0, 0, L);
    vec4 E = P(L, L, 0), F = P(0, L, 0), G = P(L, L, L), H = P(0, L, L);
    push(E, F, G); push(F, H, G);   // ceiling   :101-102
    push(C, B, A); push(C, D, B);   // floor     :105-106
    push(A, E, C); push(C, E, G);   // left      :109-110
    push(F, B, D); push(H, F, D);   // right     :113-114
    push(G, D, C); push(G, H, D);   // back      :117-118
  }
  const float blocks[2][5][2] = {
      // {Ax,Az},{Bx,Bz},{Cx,Cz},{Dx,Dz},{height,-}
      {{290, 114}, {130, 65}, {240, 272}, {82, 225}, {165, 0}},     // :123-131
      {{423, 247}, {265, 296}, {472, 406}, {314, 456}, {330, 0}}};  // :156-164
  for (int b = 0; b < 2; ++b) {
    float h = blocks[b][4][0];
    vec4 A = P(blocks[b][0][0], 0, blocks[b][0][1]), B = P(blocks[b][1][0], 0, blocks[b][1][1]);
    vec4 C = P(blocks[b][2][0], 0, blocks[b][2][1]), D = P(blocks[b][3][0], 0, blocks[b][3][1]);
    vec4 E = A, F = B, G = C, H = D;
    E.y = F.y = G.y = H.y = h;
    push(E, B, A); push(E, F, B); push(F, D, B); push(F, H, D); push(H, C, D);   // :134-151
    push(H, G, C); push(G, E, C); push(E, A, C); push(G, F, E); push(G, H, F);
  }
  const float s = 2 / L;   // :192
  for (size_t i = 0; i < v0s.size(); ++i) {
    Triangle t;
    vec4* vs[3] = {&v0s[i], &v1s[i], &v2s[i]};
    for (vec4* v : vs) {
      *v = *v * s;
      *v = *v - V4(1, 1, 1, 1);
      v->x *= -1; v->y *= -1; v->w = 1.0f;
    }
    t.v0 = v0s[i]; t.v1 = v1s[i]; t.v2 = v2s[i];
    vec3 e1 = V3(t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z);   // :29-38
    vec3 e2 = V3(t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z);
    vec3 n = normalize(cross(e2, e1));
    t.normal = V4(n, 1.0f);
    triangles.push_back(t);
  }
}

// skeleton.cpp:512-532
static void TransformationMatrix(mat4& M, vec3 theta, vec4 camera) {
  float cx = std::cos(theta.x), sx = std::sin(theta.x);
  float cy = std::cos(theta.y), sy = std::sin(theta.y);
  float cz = std::cos(theta.z), sz = std::sin(theta.z);
  M.m[0][0] = cy * cz;  M.m[0][1] = cy * sz;  M.m[0][2] = -sy;  M.m[0][3] = 0;
  M.m[1][0] = (-cx * sz) + (sx * sy * cz);
  M.m[1][1] = (cx * cz) + (sx * sy * sz);
  M.m[1][2] = sx * cy;  M.m[1][3] = 0;
  M.m[2][0] = (sx * sz) + (cx * sy * cz);
  M.m[2][1] = (-sx * cz) + (cx * sy * sz);
  M.m[2][2] = cx * cy;  M.m[2][3] = 0;
  M.m[3][0] = -camera.x; M.m[3][1] = -camera.y; M.m[3][2] = -camera.z; M.m[3][3] = 1;
}

// Tessendorf.h:47-132 — CreateSurface + UpdateHeight.
static void CreateSurface(Grid& GRID, int triangle_number, float height, double time) {
  const float HALF_W = ACTUAL_WIDTH / 2;             // :24
  const float max_wave = 0.4f, amplitude = 0.2f, gravity = 9.8f;   // :31-34
  const vec2 wind_dir{0.424f, 0.316f};
  const float step = HALF_W / 5;                     // :36
  const double e_r = 0.5, e_i = 0.5;                 // :40-41
  int width_points = triangle_number + 1;
  int total_points = (int)std::pow((double)width_points, 2);
  float triangle_size = ACTUAL_WIDTH / (float)triangle_number;
  GRID.geometric_points.assign(total_points, V4(0, 0, 0, 0));
  for (int i = 0; i < total_points; ++i) {           // :57-66
    vec4 p;
    p.x = (i % width_points) * triangle_size;
    p.z = height;
    p.y = (float)std::floor((double)(i / width_points)) * triangle_size;
    p.w = 0;
    p = p - V4(HALF_W, HALF_W, HALF_W, HALF_W);
    p.x *= 1; p.y *= 1; p.w = 1.0f;
    GRID.geometric_points[i] = p;
  }
  GRID.side_points = width_points;
  GRID.height = height;
  // UpdateHeight :77-132 (float loop counters decide the wave count)
  float multiplier = (float)(1 / std::sqrt(2.0));
  for (size_t i = 0; i < GRID.geometric_points.size(); ++i) {
    float x0x = GRID.geometric_points[i].x, x0y = GRID.geometric_points[i].y;
    std::complex<float> result = 0;
    for (float n = 0; n < HALF_W; n = n + step) {
      for (float m = 0; m < HALF_W; m = m + step) {
        float k_x = (float)((2 * PI_REF * (n - HALF_W / 2)) / max_wave);
        float k_z = (float)((2 * PI_REF * (m - HALF_W / 2)) / max_wave);
        float kk = k_x * k_x + k_z * k_z;                 // glm::dot(vec2)
        float nk = std::sqrt(kk);
        float w = std::sqrt(gravity * nk);
        float k4 = (float)std::pow((double)nk, 4);
        float kx0 = k_x * x0x + k_z * x0y;
        std::complex<float> eikx0(std::cos(kx0), std::sin(kx0));
        std::complex<float> neg_exp((float)std::cos(-time * w), (float)std::sin(-time * w));
        std::complex<float> pos_exp((float)std::cos(time * w), (float)std::sin(time * w));
        float phil_pos = 0, phil_neg = 0;
        if (k4 != 0) {
          float kw = k_x * wind_dir.x + k_z * wind_dir.y;
          float nkw = (-k_x) * wind_dir.x + (-k_z) * wind_dir.y;
          phil_pos = (float)(amplitude * (std::exp(-1 / kk) / k4) * std::pow((double)std::fabs(kw), 2));
          phil_neg = (float)(amplitude * (std::exp(-1 / kk) / k4) * std::pow((double)std::fabs(nkw), 2));
        }
        std::complex<float> sqr_p(std::sqrt(phil_pos));
        std::complex<float> sqr_n(std::sqrt(phil_neg));
        std::complex<float> random_vars((float)e_r, (float)e_i);
        std::complex<float> part_a = multiplier * random_vars * sqr_p;
        std::complex<float> part_b = multiplier * random_vars * sqr_n;
        std::complex<float> partial = ((part_a * pos_exp) + (part_b * neg_exp));
        result += partial * eikx0;
      }
    }
    GRID.geometric_points[i].z = std::abs(result);
    GRID.geometric_points[i].z += GRID.height;
  }
}

// ------------------------------------------------------ geometry kernels --
// Intersections.h:458-497
static bool ClosestIntersection(vec4 start, vec4 dir, Intersection& ci, const mat4& matrix,
                                const std::vector<Triangle>& triangles) {
  bool found = false;
  ci.distance = M_BIG;
  for (int i = 0; i < (int)triangles.size(); ++i) {
    vec4 v0 = matrix * triangles[i].v0, v1 = matrix * triangles[i].v1, v2 = matrix * triangles[i].v2;
    vec3 e1 = V3(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z);
    vec3 e2 = V3(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z);
    vec3 b = V3(start.x - v0.x, start.y - v0.y, start.z - v0.z);
    vec3 direction = V3(dir.x, dir.y, dir.z);
    mat3 A = columns(-direction, e1, e2);
    vec3 x = inverse(A) * b;
    if (x.x >= 0 && x.y >= 0 && x.z >= 0 && (x.y + x.z) < 1) {
      if (x.x < ci.distance) {
        found = true;
        ci.position = start + dir * x.x;
        ci.distance = x.x;
        ci.index = i;
      }
    }
  }
  return found;
}

// Intersections.h:81-143 — Woo's box entry test on boxes stored with flipped z.
static bool HitBoundingBox(const AABB& box, vec4 start, vec4 dir, vec4& hit) {
  enum { RIGHT = 0, LEFT = 1, MIDDLE = 2 };
  int quadrant[3];
  bool inside = true;
  float min_b[3] = {box.min.x, box.min.y, box.max.z};
  float max_b[3] = {box.max.x, box.max.y, box.min.z};
  float candidate_plane[3] = {0, 0, 0}, distances[3], position[3];
  dir = normalize(dir);
  float origin[3] = {start.x, start.y, start.z};
  float direction[3] = {dir.x, dir.y, dir.z};
  for (int i = 0; i < 3; ++i) {
    if (origin[i] < min_b[i]) { quadrant[i] = LEFT; candidate_plane[i] = min_b[i]; inside = false; }
    else if (origin[i] > max_b[i]) { quadrant[i] = RIGHT; candidate_plane[i] = max_b[i]; inside = false; }
    else quadrant[i] = MIDDLE;
  }
  if (inside) { hit = V4(start.x, start.y, start.z, 1.0f); return true; }
  for (int i = 0; i < 3; ++i) {
    if (direction[i] != 0 && quadrant[i] != MIDDLE) distances[i] = (candidate_plane[i] - origin[i]) / direction[i];
    else distances[i] = -1;
  }
  int largest = 0;
  for (int i = 0; i < 3; ++i) if (distances[largest] < distances[i]) largest = i;
  if (distances[largest] < 0) return false;
  for (int i = 0; i < 3; ++i) {
    if (largest != i) {
      position[i] = origin[i] + (distances[largest] * direction[i]);
      if (position[i] < min_b[i] || position[i] > max_b[i]) return false;
    } else position[i] = candidate_plane[i];
  }
  hit = V4(position[0], position[1], position[2], 1.0f);
  return true;
}

// Intersections.h:499-513 — Rodrigues rotation taking c_unit_dir onto w_unit_dir.
static mat3 findRotationMatrix(vec3 c_unit_dir, vec3 w_unit_dir) {
  vec3 v = cross(c_unit_dir, w_unit_dir);
  float c = dot(c_unit_dir, w_unit_dir);
  mat3 v_x;
  v_x.m[0][0] = 0;    v_x.m[1][0] = -v.z; v_x.m[2][0] = v.y;
  v_x.m[0][1] = v.z;  v_x.m[1][1] = 0;    v_x.m[2][1] = -v.x;
  v_x.m[0][2] = -v.y; v_x.m[1][2] = v.x;  v_x.m[2][2] = 0;
  return identity3() + v_x + ((v_x * v_x) * (1 / (1 + c)));
}

// Intersections.h:307-456.  second_hit is left unset by the reference on the
// paths where th2 <= 0; those paths always have valid == false, so the zero
// used here is never read.
static bool HitCylinder(vec4 start, vec4 dir, const PhotonSeg& seg, CylIntersection& isect) {
  bool first_inte = false;
  vec3 l_unit_dir = normalize(xyz(dir));
  vec4 d4 = seg.end - seg.start;
  vec3 difference = xyz(d4);
  float len = length(difference);
  vec3 w_unit_dir = normalize(V3(0.0f, len, 0.0f));
  float S_inv = 1 / len;
  vec3 c_unit_dir = difference * S_inv;
  mat3 R_inv = findRotationMatrix(c_unit_dir, w_unit_dir);
  vec3 T_inv = -((R_inv * xyz(seg.start)) * S_inv);
  vec3 origin_prime = (R_inv * (xyz(start) * S_inv)) + T_inv;
  vec3 dir_prime = R_inv * l_unit_dir;
  vec3 first_hit = V3(0, 0, 0), second_hit = V3(0, 0, 0);
  float t0 = -1, t1 = -1;
  float a = dir_prime.x * dir_prime.x + dir_prime.z * dir_prime.z;
  float b = 2 * origin_prime.x * dir_prime.x + 2 * origin_prime.z * dir_prime.z;
  float rs = seg.radius * S_inv;
  // float + float - double, rounded to float once (pow promotes to double) :344-346
  float c = (float)((double)(origin_prime.x * origin_prime.x + origin_prime.z * origin_prime.z) -
                    (double)rs * (double)rs);
  float b24ac = b * b - 4 * a * c;
  if (b24ac < 0) return false;
  float sq = sqrtf(b24ac);
  t0 = (-b + sq) / (2 * a);
  t1 = (-b - sq) / (2 * a);
  if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
  float y0 = origin_prime.y + t0 * dir_prime.y;
  float y1 = origin_prime.y + t1 * dir_prime.y;
  if (y0 < 0) {
    if (y1 < 0) return false;
    float th1 = t0 + (t1 - t0) * y0 / (y0 - y1);
    if (th1 <= 0) return false;
    first_hit = origin_prime + (dir_prime * th1);
    first_inte = true;
    if (y1 <= 1) second_hit = origin_prime + (dir_prime * t1);
    else {
      float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1);
      if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
    }
  } else if (y0 >= 0 && y0 <= 1) {
    if (t0 <= 0) return false;
    first_hit = origin_prime + (dir_prime * t0);
    first_inte = true;
    if (y1 >= 0) {
      if (y1 <= 1) second_hit = origin_prime + (dir_prime * t1);
      else {
        float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1);
        if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
      }
    } else {
      float th2 = t0 + (t1 - t0) * (y0) / (y0 - y1);
      if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
    }
  } else if (y0 > 1) {
    if (y1 > 1) return false;
    float th = t0 + (t1 - t0) * (y0 - 1) / (y0 - y1);
    if (th <= 0) return false;
    first_hit = origin_prime + (dir_prime * th);
    first_inte = true;
    if (y1 >= 0) second_hit = origin_prime + (dir_prime * t1);
    else {
      float th2 = t0 + (t1 - t0) * (y0) / (y0 - y1);
      if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
    }
  }
  if ((t0 > 0) && (t1 > 0) && (t1 > t0)) {
    isect.valid = true;
    float tb0 = length(seg.start - seg.orig_start);
    float tb_minus = tb0 + (first_hit.y / S_inv);
    float tb_plus = tb0 + (second_hit.y / S_inv);
    mat3 R = inverse(R_inv);
    isect.entry_point = (R * (first_hit - T_inv)) / S_inv;
    isect.exit_point = (R * (second_hit - T_inv)) / S_inv;
    isect.tb_minus = std::fmin(tb_minus, tb_plus);
    isect.tb_plus = std::fmax(tb_minus, tb_plus);
    isect.tc_minus = length(isect.entry_point);
    isect.tc_plus = length(isect.exit_point);
  } else isect.valid = false;
  return first_inte;
}

// Intersections.h:145-305.  The quadric is the double cone of half-angle tan_theta (the reference takes the
// cosine and sine of the tangent itself) about the y axis of the frame in which orig_start is the origin and
// orig_start -> end is the unit y axis; `min` is the y of seg.start in that frame.  pow( float, 2 ) promotes to
// double: the three coefficients are double expressions rounded to float once.  second_hit is left unset by
// the reference where th2 <= 0; the zero used here is only read when such a hit is also `valid`.
static bool HitCone(vec4 start, vec4 dir, const PhotonSeg& seg, CylIntersection& isect) {
  bool first_inte = false;
  vec3 l_unit_dir = normalize(xyz(dir));
  vec4 d4 = seg.end - seg.orig_start;
  vec3 difference = xyz(d4);
  float len = length(difference);
  vec3 w_unit_dir = normalize(V3(0.0f, len, 0.0f));
  float S_inv = 1 / len;
  vec3 c_unit_dir = difference * S_inv;
  mat3 R_inv = findRotationMatrix(c_unit_dir, w_unit_dir);
  vec3 T_inv = -((R_inv * xyz(seg.orig_start)) * S_inv);
  vec3 origin_prime = (R_inv * (xyz(start) * S_inv)) + T_inv;
  vec3 dir_prime = R_inv * l_unit_dir;
  float height = length(d4);
  float r1 = seg.radius * S_inv;
  float tan_theta = r1 / height;
  float min = (((R_inv * xyz(seg.start)) * S_inv) + T_inv).y;
  vec3 first_hit = V3(0, 0, 0), second_hit = V3(0, 0, 0);
  float t0 = -1, t1 = -1;
  const double c2 = (double)std::cos(tan_theta) * (double)std::cos(tan_theta);
  const double s2 = (double)std::sin(tan_theta) * (double)std::sin(tan_theta);
  float a = (float)(c2 * (dir_prime.x * dir_prime.x + dir_prime.z * dir_prime.z) - s2 * dir_prime.y * dir_prime.y);
  float b = (float)(c2 * (2 * origin_prime.x * dir_prime.x + 2 * origin_prime.z * dir_prime.z) -
                    s2 * 2 * origin_prime.y * dir_prime.y);
  float c = (float)(c2 * (origin_prime.x * origin_prime.x + origin_prime.z * origin_prime.z) -
                    s2 * origin_prime.y * origin_prime.y);
  float b24ac = b * b - 4 * a * c;
  if (b24ac < 0) return false;
  float sq = sqrtf(b24ac);
  t0 = (-b + sq) / (2 * a);
  t1 = (-b - sq) / (2 * a);
  if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
  float y0 = origin_prime.y + t0 * dir_prime.y;
  float y1 = origin_prime.y + t1 * dir_prime.y;
  if (y0 < min) {
    if (y1 < min) return false;
    float th1 = t0 + (t1 - t0) * (y0 - min) / (y0 - y1);
    if (th1 <= 0) return false;
    first_hit = origin_prime + (dir_prime * th1);
    first_inte = true;
    if (y1 <= 1) second_hit = origin_prime + (dir_prime * t1);
    else {
      float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1);
      if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
    }
  } else if (y0 >= min && y0 <= 1) {
    if (t0 <= 0) return false;
    first_hit = origin_prime + (dir_prime * t0);
    first_inte = true;
    if (y1 >= min) {
      if (y1 <= 1) second_hit = origin_prime + (dir_prime * t1);
      else {
        float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1);
        if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
      }
    } else {
      float th2 = t0 + (t1 - t0) * (y0 - min) / (y0 - y1);
      if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
    }
  } else if (y0 > 1) {
    if (y1 > 1) return false;
    float th = t0 + (t1 - t0) * (y0 - 1) / (y0 - y1);
    if (th <= 0) return false;
    first_hit = origin_prime + (dir_prime * th);
    first_inte = true;
    if (y1 >= 0) second_hit = origin_prime + (dir_prime * t1);       // :275 compares with 0, not min
    else {
      float th2 = t0 + (t1 - t0) * (y0 - min) / (y0 - y1);
      if (th2 > 0) second_hit = origin_prime + (dir_prime * th2);
    }
  }
  if ((t0 > 0) && (t1 > 0) && (t1 > t0)) {
    isect.valid = true;
    mat3 R = inverse(R_inv);
    isect.entry_point = (R * (first_hit - T_inv)) / S_inv;
    isect.exit_point = (R * (second_hit - T_inv)) / S_inv;
    float tb0 = length(seg.start - seg.orig_start);
    isect.tb_minus = tb0 + (first_hit.y / S_inv);
    isect.tb_plus = tb0 + (second_hit.y / S_inv);
    isect.tc_minus = length(isect.entry_point);
    isect.tc_plus = length(isect.exit_point);
  } else isect.valid = false;
  return first_inte;
}

// CastPhotons.h:1143-1146
static double Transmittance(double length, double ext) { return std::exp(-ext * length); }

// ------------------------------------------------- heterogeneous medium --
// Heterogeneous.h:137-185.  The reference indexes GRID out of bounds for
// samples outside the grid (undefined behaviour); here such indices are
// clamped into the array.  b_r = b_l always (missing braces, :158).
static float FindAverageExtinction3D(const Grid& GRID, vec3 start, vec3 direction,
                                     const std::vector<float>& distances, const std::vector<vec3>& indexes,
                                     vec3 cell_dimension, std::vector<float>& extinctions) {
  float average_extinction = 0;
  const int side = GRID.side_points;
  const int last = side * side - 1;
  auto at = [&](int i) { return GRID.geometric_points[std::min(std::max(i, 0), last)].z; };
  for (size_t i = 0; i < indexes.size(); ++i) {
    vec3 p = start + (direction * distances[i]);
    vec3 mn = cell_dimension * indexes[i];
    vec3 delta = p - mn;
    vec3 proportion = V3(std::fabs(delta.x), std::fabs(delta.y), std::fabs(delta.z)) / cell_dimension;
    int t_l = (int)((indexes[i].y * side) + indexes[i].x);
    int b_l = (int)(((indexes[i].y + 1) * side) + indexes[i].x);
    if (indexes[i].y + 1 >= side) b_l = t_l;
    int b_r = b_l + 1;
    int t_r = t_l + 1;
    if (indexes[i].x + 1 >= side) t_r = t_l;
    b_r = b_l;
    float tl = at(t_l), bl = at(b_l), tr = at(t_r), br = at(b_r);
    float left = tl + ((bl - tl) * proportion.y);
    float right = tr + ((br - tr) * proportion.y);
    float ext_i = left + ((right - left) * proportion.x);
    if (ext_i < 0) ext_i = 0;
    extinctions.push_back(ext_i);
    average_extinction += ext_i;
  }
  return average_extinction / (indexes.size());
}

// Heterogeneous.h:29-135 — 3-D DDA through the (side-1)^2 x 2 cell grid.
static float Extinction3D(const Grid& grid, vec4 start, vec4 end, std::vector<vec2>& dist_ext) {
  float t_x, t_y, t_z, t_max;
  vec3 grid_min = V3(-1, -1, 2);
  vec3 ray_direction = normalize(xyz(end) - xyz(start));
  vec3 grid_resolution = V3((float)(grid.side_points - 1), (float)(grid.side_points - 1), ACTUAL_WIDTH);
  vec3 cell = V3(ACTUAL_WIDTH, ACTUAL_WIDTH, ACTUAL_WIDTH) / grid_resolution;
  vec3 delta_t;
  vec3 o = xyz(start) - grid_min;
  t_max = length(xyz(end - start));
  auto setup = [](float d, float og, float c, float& delta, float& t) {
    if (d < 0) { delta = -c / d; t = (std::floor(og / c) * c - og) / d; }
    else       { delta = c / d;  t = ((std::floor(og / c) + 1) * c - og) / d; }
  };
  setup(ray_direction.x, o.x, cell.x, delta_t.x, t_x);
  setup(ray_direction.y, o.y, cell.y, delta_t.y, t_y);
  setup(ray_direction.z, o.z, cell.z, delta_t.z, t_z);
  float t = 0;
  vec3 q = o / cell;
  vec3 cell_index = V3(std::floor(q.x), std::floor(q.y), std::floor(q.z));
  vec3 prev_index = cell_index;
  std::vector<vec3> indexes;
  std::vector<float> distances;
  indexes.push_back(cell_index);
  distances.push_back(t);
  for (int guard = 0; guard < 1000000; ++guard) {
    if ((t_x < t_y) && (t_x < t_z)) { t = t_x; t_x += delta_t.x; cell_index.x += (ray_direction.x < 0) ? -1.f : 1.f; }
    else if (t_y < t_z)            { t = t_y; t_y += delta_t.y; cell_index.y += (ray_direction.y < 0) ? -1.f : 1.f; }
    else                           { t = t_z; t_z += delta_t.z; cell_index.z += (ray_direction.z < 0) ? -1.f : 1.f; }
    if (t > t_max || cell_index.x < 0 || cell_index.y < 0 || cell_index.x > grid_resolution.x ||
        cell_index.y > grid_resolution.y) {
      indexes.push_back(prev_index);
      distances.push_back(t_max);
      break;
    } else {
      indexes.push_back(cell_index);
      distances.push_back(t);
    }
    prev_index = cell_index;
  }
  std::vector<float> extinctions;
  float avg = FindAverageExtinction3D(grid, o, ray_direction, distances, indexes, cell, extinctions);
  for (size_t i = 0; i < indexes.size(); ++i) dist_ext.push_back(vec2{distances[i], extinctions[i]});
  return avg;
}

// -------------------------------------------------------------- integrals --
// BeamRadiance.h:86-110 — evaluated in float (std::exp(float), float products),
// widened to double only for the final quotient.
static double Integral_721(const PhotonSeg& s, const CylIntersection& i, float extinction, vec4 dir) {
  vec3 beam_dir = normalize(xyz(s.end) - xyz(s.start));
  vec3 camera_dir = normalize(xyz(dir));
  float cos_theta = dot(beam_dir, -camera_dir);
  float num_f = std::exp(-extinction * (i.tc_minus - i.tc_plus) * (std::fabs(cos_theta) - 1)) - 1;
  float den_f = std::exp(extinction * (i.tc_minus + i.tb_minus)) * extinction * (std::fabs(cos_theta) - 1);
  double numerator = num_f, denominator = den_f;
  return numerator / denominator;
}

// BeamRadiance.h:63-84 — the cone estimator: a Riemann sum in steps of 0.001 along the view ray between the two
// hits, the beam width at the matching distance along the beam growing linearly from the apex.
static double Integral_721_ada(const PhotonSeg& s, const CylIntersection& i, float extinction, float scattering_c,
                               vec4 dir) {
  vec3 beam_dir = normalize(xyz(s.end) - xyz(s.start));
  vec3 camera_dir = normalize(xyz(dir));
  double cos_theta = dot(beam_dir, -camera_dir);
  double length_const = length(s.orig_start - s.end);
  double integrand = 0;
  double dt_c = 0.001;
  for (double tc = i.tc_minus; tc < i.tc_plus; tc = tc + dt_c) {
    double tb = i.tb_minus - (std::fabs(cos_theta) * (tc - i.tc_minus));
    double current_r = (s.radius / length_const) * tb;
    double constant = Transmittance(tb, extinction);
    double transmitted = Transmittance(tc, extinction) * constant * (scattering_c / std::pow(current_r, 2)) * dt_c;
    integrand += transmitted;
  }
  return integrand;
}

// BeamRadiance.h:211-288
// het == false is Integral_73 (:112-209) with HETEROGENEOUS off: sigma_t along both paths, sigma_s as is.
static double Integral_73Het(const Grid& GRID, const PhotonSeg& s, const CylIntersection& i, float extinction_c,
                             float scattering_c, vec4 dir, bool het = true) {
  double transmitted = 0;
  dvec3 a = D3(xyz(s.start)), c = D3(i.exit_point);
  dvec3 b = D3(xyz(s.end)), d = D3(i.entry_point);
  dvec3 ab = dnormalize(b - a);
  dvec3 cd = dnormalize(d - c);
  double u_dot_u = ddot(ab, ab), u_dot_v = ddot(ab, cd), v_dot_v = ddot(cd, cd);
  double u_dot_w = ddot(ab, a - c), v_dot_w = ddot(cd, a - c);
  double denominator = u_dot_u * v_dot_v - u_dot_v * u_dot_v;
  if ((denominator != 0) && (u_dot_u != 0) && (v_dot_v != 0)) {
    double t_ab = (u_dot_v * v_dot_w - v_dot_v * u_dot_w) / denominator;
    double t_cd = (u_dot_u * v_dot_w - u_dot_v * u_dot_w) / denominator;
    dvec3 point_beam = a + (t_ab * ab);
    dvec3 point_view = c + (t_cd * cd);
    dvec3 os = D3(xyz(s.orig_start));
    double view_distance = dlength(point_view);
    double beam_distance = dlength(point_beam - os);
    bool bound1 = (view_distance >= dlength(d)) && (view_distance <= dlength(c));
    bool bound2 = (beam_distance >= dlength(a - os)) && (beam_distance <= dlength(b - os));
    if (bound1 && bound2) {
      double t_bc = dlength(point_beam - os);
      double t_cb = view_distance;
      float scattering = scattering_c;
      if (het) {
        float proportion = (float)((t_bc - dlength(a - os)) / (dlength(b - os) - dlength(a - os)));
        float additional = s.s_ext + ((s.e_ext - s.s_ext) * proportion);
        float beam_extinction = s.c_ext + ((additional - s.c_ext) / (s.seg_id + 1));
        std::vector<vec2> dist_ext;
        float view_extinction = Extinction3D(GRID, V4(0, 0, 0, 1), V4(0, 0, 0, 1) + (dir * (float)t_cb), dist_ext);
        transmitted = Transmittance(t_bc, beam_extinction) * Transmittance(t_cb, view_extinction);
        scattering = (beam_extinction / extinction_c) * scattering_c;
      } else {
        transmitted = Transmittance(t_bc, extinction_c) * Transmittance(t_cb, extinction_c);
      }
      if (transmitted < 1e-6) return 0;
      double cos_theta = ddot(ab, cd);
      double sin_theta = std::sqrt(1 - std::pow(cos_theta, 2));
      double rad = scattering / s.radius;   // float / float, then widened
      double result = transmitted * rad / sin_theta;
      return result;
    }
    return 0;
  }
  return 0;
}

// ------------------------------------------------------- random numbers --
// Philox4x32-10 (Salmon et al. 2011), the counter-based generator of the GPU
// emission kernel: draw k of photon p with seed s is the first two words of
// philox(counter = (p, k, 0, 0), key = (s_lo, s_hi)), mapped to 53 bits.
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct Rng {
  int kind = 0;                 // 0: std::mt19937 as the reference, 1: philox per photon
  std::mt19937 generator;       // CastPhotons.h:86
  uint64_t seed = 42;
  uint32_t photon = 0, draw = 0;
  double canonical() {
    if (kind == 0) { std::uniform_real_distribution<double> u(0.0, 1.0); return u(generator); }
    uint32_t c[4] = {photon, draw++, 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t hi = c[0] >> 5, lo = c[1] >> 6;     // 27 + 26 bits
    return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
  }
  double range(double a, double b) {            // std::uniform_real_distribution<double>(a, b)
    if (kind == 0) { std::uniform_real_distribution<double> u(a, b); return u(generator); }
    return canonical() * (b - a) + a;
  }
  double uniform() { return range(0.0, 1.0); }                          // CastPhotons.h:87
  double uniform_offset() { return range(-0.01, 0.01); }                // :88
  double uniform_radius() { return range(0.0, 0.05); }                  // :89
  double uniform_beam() { return range((double)0.02f, (double)0.03f); } // :90
  double uniform_small() { return range((double)0.02f, (double)0.06f); }// :91
};

// ------------------------------------------------------------- pipeline --
struct Oracle {
  Control ctl;
  std::vector<Triangle> triangles;
  mat4 matrix;
  Grid GRID;
  Rng rng;
  std::vector<PhotonBeam> beams_const, beams_scatt;
  std::vector<PhotonSeg> items;
  AABB root_aabb;
  std::vector<Node> nodes;
  std::vector<PhotonSeg> segments;        // CastPhotons.h:83 (what the tree really holds)
  std::vector<int> segment_leaf;          // node index of each entry of `segments`
  std::vector<PhotonSeg> lost_segments;   // CastPhotons.h:84
  int tree_error = 0, max_depth_limit = 4096;
  uint32_t cur_photon = 0, cur_order = 0;

  float scattering_c() const { return ctl.c.scattering_c; }
  float extinction_c() const { return ctl.extinction_c(); }

  // CastPhotons.h:1128-1141
  vec4 FindDirection(vec4 origin, vec4 centre, float /*radius*/) {
    float r1 = (float)rng.uniform();
    float r2 = (float)rng.uniform();
    float the = (float)(2 * PI_REF * r1);
    float phi = std::sqrt(r2);
    float x = centre.x + (phi * std::cos(the));
    float z = centre.z + (phi * std::sin(the));
    float y = centre.y + std::sqrt(1 - r2);
    return V4(x, y, z, 1.0f) - origin;
  }

  // CastPhotons.h:635-647
  static bool intersectPlane(vec3 n, vec3 p0, vec3 l0, vec3 l, float& t) {
    float denom = dot(n, l);
    if (std::fabs(denom) > 1e-6) {
      vec3 p0l0 = p0 - l0;
      t = dot(p0l0, n) / denom;
      return (t >= 0);
    }
    return false;
  }

  void grow(vec4& min_point, vec4& max_point, const PhotonBeam& beam) {   // CastPhotons.h:507-512
    max_point.x = std::fmax(beam.end.x, std::fmax(beam.start.x, max_point.x));
    max_point.y = std::fmax(beam.end.y, std::fmax(beam.start.y, max_point.y));
    max_point.z = std::fmin(beam.end.z, std::fmin(beam.start.z, max_point.z));
    min_point.x = std::fmin(beam.end.x, std::fmin(beam.start.x, min_point.x));
    min_point.y = std::fmin(beam.end.y, std::fmin(beam.start.y, min_point.y));
    min_point.z = std::fmax(beam.end.z, std::fmax(beam.start.z, min_point.z));
  }

  void push(std::vector<PhotonBeam>& list, PhotonBeam beam) {
    beam.photon = cur_photon; beam.order = cur_order++; beam.in_scatt = (&list == &beams_scatt);
    list.push_back(beam);
  }

  // Running mean of the extinction samples up to distance t_s. CastPhotons.h:437-443, :530-536
  static float mean_until(const std::vector<vec2>& dist_ext, float t_s) {
    float mean = 0, current_step = 0;
    while ((size_t)current_step < dist_ext.size() && dist_ext[(size_t)current_step].x < t_s) {
      mean = mean + ((dist_ext[(size_t)current_step].y - mean) / (current_step + 1));
      current_step++;
    }
    return mean;
  }

  // CastPhotons.h:392-633
  void CastBeam(int bounce, vec3 energy, vec4 origin, vec4 direction, vec4& min_point, vec4& max_point,
                std::vector<PhotonBeam>& add_to, std::vector<PhotonBeam>& other, float radius, PhotonBeam& beam) {
    const pb_control& C = ctl.c;
    bool scatt = false;
    beam.scattered = false;
    Intersection hit;
    float t_s, average_extinction;
    std::vector<vec2> dist_ext;
    if (ClosestIntersection(origin, direction, hit, matrix, triangles)) {
      t_s = length(xyz(hit.position) - xyz(origin));
      if (t_s < 0.0005) return;
    } else t_s = ACTUAL_WIDTH;
    if (C.heterogeneous) average_extinction = Extinction3D(GRID, origin, origin + (direction * t_s), dist_ext);
    else average_extinction = extinction_c();

    float scattered = (float)rng.uniform();
    if ((scattered <= (scattering_c() / extinction_c())) && C.scatter) {
      float tmp_t_s = (float)(-(std::log(1 - rng.uniform()) / average_extinction));
      if ((tmp_t_s < t_s) && (tmp_t_s > 0)) {
        t_s = tmp_t_s;
        vec4 start = origin + (t_s * normalize(direction));
        start.w = 1;
        float the = (float)(2 * PI_REF * rng.uniform());
        float phi = (float)std::acos(1 - 2 * rng.uniform());
        float x = std::sin(phi) * std::cos(the);
        float y = std::sin(phi) * std::sin(the);
        float z = std::cos(phi);
        vec3 dir = normalize(V3(x, y, z));
        vec4 dir_sample = V4(dir.x, dir.y, dir.z, 1.0f);
        float transmitted;
        if (C.heterogeneous) transmitted = (float)Transmittance(t_s, mean_until(dist_ext, t_s));
        else transmitted = (float)Transmittance(t_s, average_extinction);
        vec3 new_energy = energy * transmitted;
        if (new_energy.x == 0 && new_energy.y == 0 && new_energy.z == 0) beam.absorbed = true;
        else {
          new_energy.x = std::fmin(energy.x, std::fmax(new_energy.x, 0.f));
          new_energy.y = std::fmin(energy.y, std::fmax(new_energy.y, 0.f));
          new_energy.z = std::fmin(energy.z, std::fmax(new_energy.z, 0.f));
          PhotonBeam sc{};
          sc.scattered = true; sc.ada_width = false;
          CastBeam(bounce, new_energy, start, dir_sample, min_point, max_point, other, add_to, radius, sc);
        }
        if (C.short_beams) scatt = true;
      }
    } else {
      float tmp_t_s = (float)(-(std::log(1 - rng.uniform()) / average_extinction));
      if ((tmp_t_s < t_s) && (tmp_t_s > 0)) {
        // absorbed in flight: filled in but never pushed  :469-479
        return;
      }
    }

    if (ClosestIntersection(origin, direction, hit, matrix, triangles)) {
      beam.radius = radius;
      beam.energy = D3(energy);
      beam.index_hit = hit.index;
      beam.start = origin;
      if (scatt) { beam.end = origin + (t_s * direction); beam.end.w = 1.0f; }
      else beam.end = hit.position;
      float absorbed = (float)rng.uniform();
      beam.absorbed = (absorbed < C.absorbed);
      push(add_to, beam);
      grow(min_point, max_point, beam);
      if ((bounce < C.bounces) && (!beam.absorbed) && (!scatt)) {
        int num = bounce + 1;
        vec4 normal = triangles[hit.index].normal;
        vec4 incident = hit.position + direction;
        vec3 refl = reflect(xyz(incident), xyz(normal));
        vec4 reflected = V4(normalize(refl), 1.0f);
        vec4 bounce_dir = FindDirection(hit.position, hit.position + (reflected * 0.5f), 0);
        vec4 new_direction = V4(normalize(xyz(bounce_dir)), 1.0f);
        if (scattered > 0.8) new_direction = reflected;
        float transmitted;
        if (C.heterogeneous) transmitted = (float)Transmittance(t_s, mean_until(dist_ext, t_s));
        else transmitted = (float)Transmittance(t_s, average_extinction);
        vec3 new_energy = energy * transmitted;
        if (!(new_energy.x == 0 && new_energy.y == 0 && new_energy.z == 0)) {
          new_energy.x = std::fmin(energy.x, std::fmax(new_energy.x, 0.f));
          new_energy.y = std::fmin(energy.y, std::fmax(new_energy.y, 0.f));
          new_energy.z = std::fmin(energy.z, std::fmax(new_energy.z, 0.f));
          PhotonBeam bounced{};
          bounced.ada_width = false;
          CastBeam(num, new_energy, hit.position + (normal * 0.0001f), new_direction, min_point, max_point,
                   other, add_to, radius, bounced);
        }
      }
    } else {
      // open front of the box: clip against four planes through the camera  :558-618
      vec4 top_left = V4(-1, -1, -1, 1), top_right = V4(1, -1, -1, 1);
      vec4 bot_left = V4(-1, 1, -1, 1), bot_right = V4(1, 1, -1, 1);
      vec4 converge = matrix * V4(0, 0, C.camera[2], 1.0f);
      vec4 top = matrix * ((top_left + top_right) / 2.0f);
      vec4 right = matrix * ((top_right + bot_right) / 2.0f);
      vec4 bot = matrix * ((bot_right + bot_left) / 2.0f);
      vec4 left = matrix * ((bot_left + top_left) / 2.0f);
      vec3 p_0_0 = xyz((top + converge) / 2.0f), p_0_1 = xyz((right + converge) / 2.0f);
      vec3 p_0_2 = xyz((bot + converge) / 2.0f), p_0_3 = xyz((left + converge) / 2.0f);
      vec3 n_0_0 = normalize(cross(p_0_0 - xyz(top), xyz((matrix * top_left) - top)));
      vec3 n_0_1 = normalize(cross(p_0_1 - xyz(right), xyz((matrix * top_right) - right)));
      vec3 n_0_2 = normalize(cross(p_0_2 - xyz(bot), xyz((matrix * bot_right) - bot)));
      vec3 n_0_3 = normalize(cross(p_0_3 - xyz(left), xyz((matrix * bot_left) - left)));
      vec3 dir = normalize(xyz(direction));
      vec3 start = xyz(origin);
      float t;
      beam.absorbed = false;
      beam.radius = radius;
      beam.energy = D3(energy);
      beam.start = origin;
      beam.index_hit = -1;
      float t_min = M_BIG;
      if (intersectPlane(n_0_0, p_0_0, start, dir, t)) if (t < t_min) t_min = t;
      if (intersectPlane(n_0_1, p_0_1, start, dir, t)) if (t < t_min) t_min = t;
      if (intersectPlane(n_0_2, p_0_2, start, dir, t)) if (t < t_min) t_min = t;
      if (intersectPlane(n_0_3, p_0_3, start, dir, t)) if (t < t_min) t_min = t;
      if (t_min == M_BIG) return;
      if (scatt) { beam.end = origin + (t_s * direction); beam.end.w = 1.0f; }
      else beam.end = V4(start + (t_min * dir), 1.0f);
      push(add_to, beam);
      grow(min_point, max_point, beam);
    }
  }

  // CastPhotons.h:312-390
  AABB CastPhotonBeams(int number) {
    const pb_control& C = ctl.c;
    vec4 min_point = V4(M_BIG, M_BIG, -M_BIG, 1), max_point = V4(-M_BIG, -M_BIG, M_BIG, 1);
    vec4 light = V4(C.light_position[0], C.light_position[1], C.light_position[2], C.light_position[3]);
    vec4 origin = matrix * light;
    vec4 centre = V4(origin.x, (float)(origin.y + 1.5), origin.z, 1.0f);
    float radius = 0.05f;
    vec3 energy = V3(C.light_power[0], C.light_power[1], C.light_power[2]);
    for (int i = 0; i < number; ++i) {
      cur_photon = (uint32_t)i; cur_order = 0;
      rng.photon = (uint32_t)i; rng.draw = 0;
      PhotonBeam beam{};
      vec4 direction = FindDirection(origin, centre, radius);
      direction = normalize(direction);
      beam.ada_width = C.adaptive != 0;
      float r = (float)rng.uniform_beam();
      float r_small = (float)rng.uniform_small();
      if (C.fixed_radius) { r = C.radius; r_small = r; }
      vec3 w_u = normalize(V3(direction.x + r_small, direction.y, direction.z));
      vec3 w_v = normalize(V3(direction.x, direction.y, direction.z - r_small));
      beam.omega_u = V4(w_u, 1.0f);
      beam.omega_v = V4(w_v, 1.0f);
      float t = (float)(2 * PI_REF * rng.uniform_radius());
      double u1 = rng.uniform_radius();
      double u2 = rng.uniform_radius();
      float u = (float)(u1 + u2);
      float w_width = (u > 1) ? 2 - u : u;
      float x_offset = w_width * std::cos(t);
      float y_offset = (float)(rng.uniform_offset() * 0.1);
      float z_offset = w_width * std::sin(t);
      vec4 start = origin + V4(x_offset, y_offset, z_offset, 0);
      CastBeam(0, energy, start, direction, min_point, max_point, beams_const, beams_scatt, r, beam);
    }
    AABB root;
    root.min = min_point; root.max = max_point;
    root.mid = (min_point + max_point) / 2.0f;
    return root;
  }

  // CastPhotons.h:290-310.  The reference loops `i < beams_const.size()` over the very vector CastBeam
  // appends to, so it re-casts its own output and does not terminate; here only the beams present on
  // entry are re-cast.  As there, the re-cast copies are discarded and the scattered list is rebuilt.
  AABB RecastPhotonBeams() {
    vec4 min_point = V4(M_BIG, M_BIG, -M_BIG, 1), max_point = V4(-M_BIG, -M_BIG, M_BIG, 1);
    std::vector<PhotonBeam> copy = beams_const;
    const size_t n0 = copy.size();
    beams_scatt.clear();
    for (size_t i = 0; i < n0; ++i) {
      PhotonBeam beam = copy[i];
      cur_photon = (uint32_t)i; cur_order = 0;
      rng.photon = (uint32_t)i; rng.draw = 0;
      vec4 direction = V4(normalize(xyz(beam.end) - xyz(beam.start)), 1.0f);
      vec3 energy = V3((float)beam.energy.x, (float)beam.energy.y, (float)beam.energy.z);
      CastBeam(0, energy, beam.start, direction, min_point, max_point, copy, beams_scatt, beam.radius, beam);
    }
    AABB root;
    root.min = min_point; root.max = max_point;
    root.mid = (min_point + max_point) / 2.0f;
    return root;
  }

  // CastPhotons.h:144-288
  void BoundPhotonBeams(std::vector<PhotonBeam>& beams) {
    const pb_control& C = ctl.c;
    vec3 x_dir = normalize(V3(1, 0, 0)), y_dir = normalize(V3(0, 1, 0)), z_dir = normalize(V3(0, 0, 1));
    for (int i = 0; i < (int)beams.size(); ++i) {
      const PhotonBeam& b = beams[i];
      vec4 start = b.start, end = b.end;
      vec3 diff = xyz(end) - xyz(start);
      vec3 dir = normalize(diff);
      float len = length(diff);
      float j = 0;
      float theta_x = std::fmin(std::acos(dot(dir, x_dir)), std::acos(dot(dir, -x_dir)));
      float theta_y = std::fmin(std::acos(dot(dir, y_dir)), std::acos(dot(dir, -y_dir)));
      float theta_z = std::fmin(std::acos(dot(dir, z_dir)), std::acos(dot(dir, -z_dir)));
      float step = len;
      std::vector<vec2> dist_ext;
      int current_step = 1;
      float mean = extinction_c();
      if (C.heterogeneous) {
        Extinction3D(GRID, start, end, dist_ext);
        mean = dist_ext[current_step - 1].y;
      }
      vec3 prior = V3(start.x, start.y, start.z);
      while (j < len && len > 1e-6) {
        if (C.heterogeneous) step = dist_ext[current_step].x - dist_ext[current_step - 1].x;
        PhotonSeg s{};
        vec3 next = prior + (dir * step);
        s.start = V4(prior.x, prior.y, prior.z, 1.0f);
        if (C.heterogeneous) mean = mean + ((dist_ext[current_step].y - mean) / (current_step + 1));
        else mean = extinction_c();
        s.c_ext = mean;
        s.seg_id = current_step;
        if (len <= length(next - xyz(start))) { j = std::fabs(len); s.end = V4(end.x, end.y, end.z, 1.0f); }
        else { j = length(next - xyz(start)); s.end = V4(next.x, next.y, next.z, 1.0f); }
        prior = xyz(s.end);
        s.orig_start = start; s.orig_start.w = 1.0f;
        if (b.ada_width) {
          vec3 omega_u = normalize(xyz(b.omega_u));
          float cos_theta = dot(dir, omega_u);
          float hyp_length = j / cos_theta;
          vec4 end_u = start + (b.omega_u * hyp_length);
          s.radius = length(xyz(s.end) - xyz(end_u));
          s.ada_width = true;
        } else { s.radius = b.radius; s.ada_width = false; }
        s.scattered = b.scattered;
        s.id = i;
        s.min = V4(std::fmin(s.start.x, s.end.x), std::fmin(s.start.y, s.end.y), std::fmax(s.start.z, s.end.z), 1.0f);
        s.max = V4(std::fmax(s.start.x, s.end.x), std::fmax(s.start.y, s.end.y), std::fmin(s.start.z, s.end.z), 1.0f);
        if (len > 0.001) {
          float th; int axis;
          if (theta_x < theta_y && theta_x < theta_z) { th = theta_x; axis = 0; }
          else if (theta_y < theta_z) { th = theta_y; axis = 1; }
          else { th = theta_z; axis = 2; }
          float hypotenuse = s.radius / std::sin(th);
          float adj = s.radius / std::cos(th);
          float height = adj * std::sin(th);
          float px = (axis == 0) ? height : hypotenuse;
          float py = (axis == 1) ? height : hypotenuse;
          float pz = (axis == 2) ? height : hypotenuse;
          s.min = V4(s.min.x - px, s.min.y - py, s.min.z + pz, 1.0f);
          s.max = V4(s.max.x + px, s.max.y + py, s.max.z - pz, 1.0f);
        }
        s.mid = (s.min + s.max) / 2.0f;
        if (C.heterogeneous) {
          s.s_ext = dist_ext[current_step - 1].y;
          s.e_ext = dist_ext[current_step].y;
          if (current_step == (int)dist_ext.size() - 1) j = len + 1;
          current_step++;
        }
        items.push_back(s);
      }
    }
  }

  // ---- BuildTree, CastPhotons.h:800-1126 ----
  int newNode(const AABB& box, int depth) {             // CastPhotons.h:58-76
    Node n; n.aabb = box; n.seg[0] = n.seg[1] = -1; n.leaf = false; n.left = n.right = -1; n.depth = depth;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
  }
  static float comp(const vec4& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

  // One partition sweep over `child` along `axis`.  `greater_left` is the z
  // convention of the first-level sweeps (mid.z > node mid goes left);
  // `keep_all_ties` is the last-resort form whose ties always become leftovers.
  static void sweep(const std::vector<PhotonSeg>& child, const vec4& mid, int axis, bool greater_left,
                    bool keep_all_ties, std::vector<PhotonSeg>& l, std::vector<PhotonSeg>& r,
                    std::vector<PhotonSeg>& leftovers, bool& same) {
    float m = comp(mid, axis);
    for (const PhotonSeg& box : child) {
      float k = comp(box.mid, axis);
      bool to_l = greater_left ? (k > m) : (k < m);
      bool to_r = greater_left ? (k < m) : (k > m);
      if (to_l) { l.push_back(box); same = false; }
      else if (to_r) { r.push_back(box); same = false; }
      else if (keep_all_ties || !same) leftovers.push_back(box);
    }
  }

  void BuildTree(int parent, std::vector<PhotonSeg>& child, int leftover) {
    if (tree_error) return;
    if (nodes[parent].depth > max_depth_limit) { tree_error = 1; return; }
    if (child.size() <= 2) {
      nodes[parent].leaf = true;
      for (size_t i = 0; i < child.size(); ++i) {
        nodes[parent].seg[i] = (int)segments.size();
        segments.push_back(child[i]);
        segment_leaf.push_back(parent);
      }
      return;
    }
    nodes[parent].leaf = false;
    vec4 mx = nodes[parent].aabb.max, mn = nodes[parent].aabb.min;
    vec4 diff = V4(std::fabs(mx.x - mn.x), std::fabs(mx.y - mn.y), std::fabs(mx.z - mn.z), std::fabs(mx.w - mn.w));
    vec4 mid = nodes[parent].aabb.mid;
    std::vector<PhotonSeg> l, r, leftovers;
    bool same = true;
    auto many_ties = [&]() { return leftovers.size() > ((l.size() + r.size()) / 2); };
    // first sweep along the longest axis of the node box; z uses the flipped sense  :825-870
    int dim0 = (diff.x > diff.y && diff.x > diff.z) ? 0 : ((diff.y > diff.z) ? 1 : 2);
    sweep(child, mid, dim0, dim0 == 2, false, l, r, leftovers, same);
    if (many_ties()) same = true;
    if (same) {
      // second and third choice axes  :873-1063 (leftovers are NOT cleared in between)
      l.clear(); r.clear();
      int a = (dim0 == 0) ? 1 : 0, b = (dim0 == 2) ? 1 : 2;       // the other two axes, a < b
      int second = (comp(diff, a) > comp(diff, b)) ? a : b;
      int third = (second == a) ? b : a;
      sweep(child, mid, second, second == 2, false, l, r, leftovers, same);
      if (many_ties()) same = true;
      if (same) {
        l.clear(); r.clear();
        sweep(child, mid, third, false, true, l, r, leftovers, same);   // z here uses '<' for left
      }
    }
    if (!ctl.c.heterogeneous || (leftover > (int)leftovers.size())) {   // :1065-1077
      for (size_t i = 0; i < leftovers.size(); ++i) {
        if (rng.uniform() < 0.5) l.push_back(leftovers[i]); else r.push_back(leftovers[i]);
      }
    } else {
      for (size_t i = 0; i < leftovers.size(); ++i) lost_segments.push_back(leftovers[i]);
    }
    auto bound = [](const std::vector<PhotonSeg>& v) {   // :1079-1094 (z flipped)
      vec4 mn = V4(M_BIG, M_BIG, -M_BIG, 1), mx = V4(-M_BIG, -M_BIG, M_BIG, 1);
      for (const PhotonSeg& s : v) {
        mx.x = std::fmax(s.min.x, std::fmax(s.max.x, mx.x));
        mx.y = std::fmax(s.min.y, std::fmax(s.max.y, mx.y));
        mx.z = std::fmin(s.min.z, std::fmin(s.max.z, mx.z));
        mn.x = std::fmin(s.min.x, std::fmin(s.max.x, mn.x));
        mn.y = std::fmin(s.min.y, std::fmin(s.max.y, mn.y));
        mn.z = std::fmax(s.min.z, std::fmax(s.max.z, mn.z));
      }
      AABB box;
      box.min = V4(mn.x, mn.y, mn.z, 1.0f); box.max = V4(mx.x, mx.y, mx.z, 1.0f);
      box.mid = (box.min + box.max) / 2.0f;
      return box;
    };
    int n_left = (int)leftovers.size();
    if (!l.empty()) {
      int c = newNode(bound(l), nodes[parent].depth + 1);
      nodes[parent].left = c;
      BuildTree(c, l, n_left);
    }
    if (!r.empty()) {
      int c = newNode(bound(r), nodes[parent].depth + 1);
      nodes[parent].right = c;
      BuildTree(c, r, n_left);
    }
  }
};

// -------------------------------------------------------------- rendering --
struct RayCtx {
  vec4 start, dir;
  float max_distance;
};

static inline bool box_gate(const AABB& box, const RayCtx& r) {      // BeamRadiance.h:428-430
  vec4 hit;
  if (!HitBoundingBox(box, r.start, r.dir, hit)) return false;
  float hit_distance = length(hit - r.start);
  return hit_distance <= r.max_distance;
}

struct Medium { float extinction_c, scattering_c; const Grid* grid; bool one_dimensional = false; };

// Leaf body of BeamRadianceStandard (BeamRadiance.h:436-456) / BVHBeamRadianceHet (:378-398).
static inline void leaf_segment(const PhotonSeg& seg, const dvec3& energy, const RayCtx& r, int mode,
                                const Medium& med, dvec3& current, orc_counters& cnt) {
  CylIntersection isect;
  isect.valid = false;
  cnt.cyl_calls++;
  const double phase_f = 1 / (4 * PI_REF);
  if (mode == PB_MODE_GENERIC && seg.ada_width) {                  // BeamRadiance.h:488-507
    if (HitCone(r.start, r.dir, seg, isect)) {
      cnt.cyl_true++;
      if (isect.valid) {
        cnt.valid_hits++;
        double _int = Integral_721_ada(seg, isect, med.extinction_c, med.scattering_c, r.dir);
        current = current + (energy * phase_f) * _int;
      }
    }
    return;
  }
  if (mode == PB_MODE_GENERIC) mode = med.one_dimensional ? PB_MODE_73 : PB_MODE_721;   // :508-541
  if (HitCylinder(r.start, r.dir, seg, isect)) {
    cnt.cyl_true++;
    if (isect.valid) {
      cnt.valid_hits++;
      if (mode == PB_MODE_721) {
        double _int = Integral_721(seg, isect, med.extinction_c, r.dir);
        double rad = med.scattering_c / (PI_REF * ((double)seg.radius * (double)seg.radius));
        current = current + (energy * phase_f) * rad * _int;
      } else {
        double _int = Integral_73Het(*med.grid, seg, isect, med.extinction_c, med.scattering_c, r.dir,
                                     mode == PB_MODE_73HET);
        current = current + (energy * phase_f) * _int;
      }
    }
  }
}

// The draw SHORT_VIEW makes per camera ray (skeleton.cpp:443).  kind 0: the reference's generator, drawn for a
// whole frame in the reference's loop order (x outer, y inner, then the sub-samples) and looked up; kind 1:
// the device's draw, philox keyed by the sub-pixel.
struct ViewDraws {
  int kind = 1;
  uint32_t seed = 42;
  int sub_h = 0;                       // SCREEN_HEIGHT * SSAA
  std::vector<double> table;           // kind 0: [sub-pixel x * sub_h + sub-pixel y], NaN where no draw was made
  double draw(int px, int py) const {
    if (kind == 0) return table[(size_t)px * sub_h + py];
    uint32_t c[4] = {(uint32_t)px, (uint32_t)py, 0u, 0u};
    philox4x32_10(c, seed, 0x56494557u);
    uint64_t hi = c[0] >> 5, lo = c[1] >> 6;
    return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
  }
};

// Per-pixel setup of DrawStandard/DrawHet (skeleton.cpp:342-350) and Draw (:428-449).
static inline bool camera_ray(const Control& ctl, const std::vector<Triangle>& tris, const mat4& matrix,
                              int px, int py, RayCtx& r, const ViewDraws* view = nullptr) {
  const pb_control& C = ctl.c;
  float x_dir = px - ((C.screen_width * C.ssaa) / (float)2);
  float y_dir = py - ((C.screen_height * C.ssaa) / (float)2);
  vec4 direction = normalize(V4(x_dir, y_dir, ctl.focal(), 1.0f));
  r.start = V4(0, 0, 0, 1);
  r.dir = direction;
  Intersection c_i;
  if (!ClosestIntersection(r.start, direction, c_i, matrix, tris)) return false;
  if (C.short_view && view) {                            // skeleton.cpp:440-449
    float max_distance = length(c_i.position - r.start);
    float t_a = (float)(-(std::log(1 - view->draw(px, py)) / ctl.extinction_c()));
    if (t_a > 0) {
      max_distance = std::fmin(t_a, max_distance);
      c_i.position = r.start + (direction * max_distance);
    }
  }
  r.max_distance = length(c_i.position - r.start);      // BeamRadiance.h:426
  return true;
}

static PhotonSeg seg_from_record(const pb_segment& s) {
  PhotonSeg p{};
  p.start = V4(s.start[0], s.start[1], s.start[2], 1.0f);
  p.end = V4(s.end[0], s.end[1], s.end[2], 1.0f);
  p.orig_start = V4(s.orig_start[0], s.orig_start[1], s.orig_start[2], 1.0f);
  p.radius = s.radius; p.id = s.id; p.seg_id = s.seg_id;
  p.s_ext = s.s_ext; p.e_ext = s.e_ext; p.c_ext = s.c_ext;
  p.ada_width = s.flags & 1u; p.scattered = (s.flags >> 1) & 1u;
  return p;
}
static AABB box_from_lohi(const float lo[3], const float hi[3]) {
  AABB b;   // stored with the reference's flipped z: min.z holds the larger z
  b.min = V4(lo[0], lo[1], hi[2], 1.0f);
  b.max = V4(hi[0], hi[1], lo[2], 1.0f);
  b.mid = (b.min + b.max) / 2.0f;
  return b;
}

}  // namespace orc

using namespace orc;

struct orc_ctx { Oracle o; };

// ================================================================= C API ==
extern "C" {

int orc_load_test_model(float* tri, float* normals, int capacity) {
  std::vector<Triangle> t;
  LoadTestModel(t);
  if ((int)t.size() > capacity) return -1;
  for (size_t i = 0; i < t.size(); ++i) {
    const vec4* vs[3] = {&t[i].v0, &t[i].v1, &t[i].v2};
    for (int k = 0; k < 3; ++k) {
      tri[i * 12 + k * 4 + 0] = vs[k]->x; tri[i * 12 + k * 4 + 1] = vs[k]->y;
      tri[i * 12 + k * 4 + 2] = vs[k]->z; tri[i * 12 + k * 4 + 3] = vs[k]->w;
    }
    if (normals) { normals[i * 4] = t[i].normal.x; normals[i * 4 + 1] = t[i].normal.y;
                   normals[i * 4 + 2] = t[i].normal.z; normals[i * 4 + 3] = t[i].normal.w; }
  }
  return (int)t.size();
}

void orc_transformation_matrix(const float theta[3], const float camera[4], float m16[16]) {
  mat4 M;
  TransformationMatrix(M, V3(theta[0], theta[1], theta[2]), V4(camera[0], camera[1], camera[2], camera[3]));
  std::memcpy(m16, M.m, sizeof(float) * 16);
}

static std::vector<Triangle> tris_from_xyzw(const float* t, int n) {
  std::vector<Triangle> v(n);
  for (int i = 0; i < n; ++i) {
    v[i].v0 = V4(t[i * 12 + 0], t[i * 12 + 1], t[i * 12 + 2], t[i * 12 + 3]);
    v[i].v1 = V4(t[i * 12 + 4], t[i * 12 + 5], t[i * 12 + 6], t[i * 12 + 7]);
    v[i].v2 = V4(t[i * 12 + 8], t[i * 12 + 9], t[i * 12 + 10], t[i * 12 + 11]);
    vec3 e1 = xyz(v[i].v1) - xyz(v[i].v0), e2 = xyz(v[i].v2) - xyz(v[i].v0);
    v[i].normal = V4(normalize(cross(e2, e1)), 1.0f);
  }
  return v;
}
static std::vector<Triangle> tris_from_xyz(const float* t, int n) {
  std::vector<Triangle> v(n);
  for (int i = 0; i < n; ++i) {
    v[i].v0 = V4(t[i * 9 + 0], t[i * 9 + 1], t[i * 9 + 2], 1.0f);
    v[i].v1 = V4(t[i * 9 + 3], t[i * 9 + 4], t[i * 9 + 5], 1.0f);
    v[i].v2 = V4(t[i * 9 + 6], t[i * 9 + 7], t[i * 9 + 8], 1.0f);
    vec3 e1 = xyz(v[i].v1) - xyz(v[i].v0), e2 = xyz(v[i].v2) - xyz(v[i].v0);
    v[i].normal = V4(normalize(cross(e2, e1)), 1.0f);
  }
  return v;
}

int orc_closest_intersection(const float start[4], const float dir[4], const float m16[16], const float* tri,
                             int n_tri, float position[4], float* distance, int* index) {
  mat4 M; std::memcpy(M.m, m16, sizeof(float) * 16);
  std::vector<Triangle> t = tris_from_xyzw(tri, n_tri);
  Intersection ci; ci.position = V4(0, 0, 0, 0); ci.index = -1;
  bool f = ClosestIntersection(V4(start[0], start[1], start[2], start[3]), V4(dir[0], dir[1], dir[2], dir[3]), ci, M, t);
  position[0] = ci.position.x; position[1] = ci.position.y; position[2] = ci.position.z; position[3] = ci.position.w;
  *distance = ci.distance; *index = ci.index;
  return f ? 1 : 0;
}

int orc_hit_bounding_box(const float bmin[4], const float bmax[4], const float start[4], const float dir[4], float hit[4]) {
  AABB b; b.min = V4(bmin[0], bmin[1], bmin[2], bmin[3]); b.max = V4(bmax[0], bmax[1], bmax[2], bmax[3]);
  b.mid = (b.min + b.max) / 2.0f;
  vec4 h = V4(0, 0, 0, 0);
  bool r = HitBoundingBox(b, V4(start[0], start[1], start[2], start[3]), V4(dir[0], dir[1], dir[2], dir[3]), h);
  hit[0] = h.x; hit[1] = h.y; hit[2] = h.z; hit[3] = h.w;
  return r ? 1 : 0;
}

static void hit_quadric(bool cone, const float start[4], const float dir[4], const float s0[4], const float s1[4],
                        const float so[4], float radius, float out[12]) {
  PhotonSeg seg{};
  seg.start = V4(s0[0], s0[1], s0[2], s0[3]); seg.end = V4(s1[0], s1[1], s1[2], s1[3]);
  seg.orig_start = V4(so[0], so[1], so[2], so[3]); seg.radius = radius;
  CylIntersection ci{};
  const vec4 st = V4(start[0], start[1], start[2], start[3]), d = V4(dir[0], dir[1], dir[2], dir[3]);
  bool r = cone ? HitCone(st, d, seg, ci) : HitCylinder(st, d, seg, ci);
  for (int i = 0; i < 12; ++i) out[i] = 0;
  out[0] = r ? 1.f : 0.f;
  if (r) {
    out[1] = ci.valid ? 1.f : 0.f;
    if (ci.valid) {
      out[2] = ci.tb_minus; out[3] = ci.tb_plus; out[4] = ci.tc_minus; out[5] = ci.tc_plus;
      out[6] = ci.entry_point.x; out[7] = ci.entry_point.y; out[8] = ci.entry_point.z;
      out[9] = ci.exit_point.x; out[10] = ci.exit_point.y; out[11] = ci.exit_point.z;
    }
  }
}

void orc_hit_cylinder(const float start[4], const float dir[4], const float s0[4], const float s1[4],
                      const float so[4], float radius, float out[12]) {
  hit_quadric(false, start, dir, s0, s1, so, radius, out);
}

void orc_hit_cone(const float start[4], const float dir[4], const float s0[4], const float s1[4],
                  const float so[4], float radius, float out[12]) {
  hit_quadric(true, start, dir, s0, s1, so, radius, out);
}

double orc_integral_721_ada(const float s0[4], const float s1[4], const float so[4], float radius, float tb_minus,
                            float tc_minus, float tc_plus, float extinction, float scattering, const float dir[4]) {
  PhotonSeg seg{};
  seg.start = V4(s0[0], s0[1], s0[2], s0[3]); seg.end = V4(s1[0], s1[1], s1[2], s1[3]);
  seg.orig_start = V4(so[0], so[1], so[2], so[3]); seg.radius = radius;
  CylIntersection ci{};
  ci.tb_minus = tb_minus; ci.tc_minus = tc_minus; ci.tc_plus = tc_plus;
  return Integral_721_ada(seg, ci, extinction, scattering, V4(dir[0], dir[1], dir[2], dir[3]));
}

void orc_find_rotation_matrix(const float c[3], const float w[3], float m9[9]) {
  mat3 R = findRotationMatrix(V3(c[0], c[1], c[2]), V3(w[0], w[1], w[2]));
  std::memcpy(m9, R.m, sizeof(float) * 9);
}

double orc_integral_721(const float s0[4], const float s1[4], float tb_minus, float tc_minus, float tc_plus,
                        float extinction, const float dir[4]) {
  PhotonSeg seg{};
  seg.start = V4(s0[0], s0[1], s0[2], s0[3]); seg.end = V4(s1[0], s1[1], s1[2], s1[3]);
  CylIntersection ci{};
  ci.tb_minus = tb_minus; ci.tc_minus = tc_minus; ci.tc_plus = tc_plus;
  return Integral_721(seg, ci, extinction, V4(dir[0], dir[1], dir[2], dir[3]));
}

int orc_create_surface(int triangle_number, float height, double time, float* pts) {
  Grid g;
  CreateSurface(g, triangle_number, height, time);
  for (size_t i = 0; i < g.geometric_points.size(); ++i) {
    pts[i * 4] = g.geometric_points[i].x; pts[i * 4 + 1] = g.geometric_points[i].y;
    pts[i * 4 + 2] = g.geometric_points[i].z; pts[i * 4 + 3] = g.geometric_points[i].w;
  }
  return (int)g.geometric_points.size();
}

static Grid grid_from_sigma(const float* sigma, int side) {
  Grid g;
  g.side_points = side;
  g.geometric_points.resize((size_t)side * side);
  float tri = ACTUAL_WIDTH / (float)(side - 1);
  for (int i = 0; i < side * side; ++i)
    g.geometric_points[i] = V4((i % side) * tri - 1, (float)(i / side) * tri - 1, sigma[i], 1.0f);
  return g;
}

int orc_extinction3d(const float* sigma, int side, const float start[4], const float end[4], float* dist_ext,
                     int capacity, float* mean) {
  Grid g = grid_from_sigma(sigma, side);
  std::vector<vec2> de;
  *mean = Extinction3D(g, V4(start[0], start[1], start[2], start[3]), V4(end[0], end[1], end[2], end[3]), de);
  int n = (int)std::min<size_t>(de.size(), (size_t)capacity);
  for (int i = 0; i < n; ++i) { dist_ext[2 * i] = de[i].x; dist_ext[2 * i + 1] = de[i].y; }
  return (int)de.size();
}

double orc_integral_73het(const float* sigma, int side, const pb_segment* s, const float entry[3],
                          const float exit_[3], float extinction_c, float scattering_c, const float dir[4]) {
  Grid g = grid_from_sigma(sigma, side);
  PhotonSeg seg = seg_from_record(*s);
  CylIntersection ci{};
  ci.valid = true;
  ci.entry_point = V3(entry[0], entry[1], entry[2]);
  ci.exit_point = V3(exit_[0], exit_[1], exit_[2]);
  return Integral_73Het(g, seg, ci, extinction_c, scattering_c, V4(dir[0], dir[1], dir[2], dir[3]));
}

// SDLauxiliary.h:149-161
void orc_put_pixel(const double rgb[3], uint32_t* argb) {
  float c[3] = {(float)rgb[0], (float)rgb[1], (float)rgb[2]};
  uint32_t q[3];
  for (int i = 0; i < 3; ++i) {
    float v = 255 * c[i];
    v = std::min(std::max(v, 0.f), 255.f);
    q[i] = (uint32_t)v;
  }
  *argb = (128u << 24) + (q[0] << 16) + (q[1] << 8) + q[2];
}

orc_ctx* orc_create(const pb_control* control) {
  orc_ctx* c = new orc_ctx();
  c->o.ctl.c = *control;
  LoadTestModel(c->o.triangles);
  TransformationMatrix(c->o.matrix, V3(control->theta[0], control->theta[1], control->theta[2]),
                       V4(control->camera[0], control->camera[1], control->camera[2], control->camera[3]));
  c->o.rng.generator.seed(control->seed);
  c->o.rng.seed = control->seed;
  return c;
}
void orc_destroy(orc_ctx* c) { delete c; }

int orc_cast_photon_beams(orc_ctx* c, int number, int rng_kind) {
  Oracle& o = c->o;
  o.rng.kind = rng_kind;
  o.beams_const.clear(); o.beams_scatt.clear();
  o.root_aabb = o.CastPhotonBeams(number);
  return (int)(o.beams_const.size() + o.beams_scatt.size());
}
// RecastPhotonBeams (terminating form, see above) with the random stream re-keyed by `seed`.
int orc_recast_photon_beams(orc_ctx* c, int rng_kind, uint64_t seed) {
  Oracle& o = c->o;
  o.rng.kind = rng_kind;
  o.rng.seed = seed;
  if (rng_kind == 0) o.rng.generator.seed(seed);
  o.root_aabb = o.RecastPhotonBeams();
  return (int)o.beams_scatt.size();
}
int orc_beam_counts(orc_ctx* c, int* n_const, int* n_scatt) {
  *n_const = (int)c->o.beams_const.size(); *n_scatt = (int)c->o.beams_scatt.size();
  return 0;
}
int orc_get_beams(orc_ctx* c, pb_beam* out, int capacity) {
  Oracle& o = c->o;
  int n = 0;
  for (const std::vector<PhotonBeam>* list : {&o.beams_const, &o.beams_scatt})
    for (const PhotonBeam& b : *list) {
      if (n >= capacity) return -1;
      pb_beam& d = out[n++];
      std::memset(&d, 0, sizeof(d));
      d.start[0] = b.start.x; d.start[1] = b.start.y; d.start[2] = b.start.z; d.start[3] = b.start.w;
      d.end[0] = b.end.x; d.end[1] = b.end.y; d.end[2] = b.end.z; d.end[3] = b.end.w;
      d.omega_u[0] = b.omega_u.x; d.omega_u[1] = b.omega_u.y; d.omega_u[2] = b.omega_u.z; d.omega_u[3] = b.omega_u.w;
      d.radius = b.radius;
      d.energy[0] = (float)b.energy.x; d.energy[1] = (float)b.energy.y; d.energy[2] = (float)b.energy.z;
      d.index_hit = b.index_hit;
      d.flags = (b.ada_width ? 1u : 0u) | (b.scattered ? 2u : 0u) | (b.absorbed ? 4u : 0u) | (b.in_scatt ? 8u : 0u);
      d.photon = b.photon; d.order = b.order;
    }
  return n;
}
int orc_bound_photon_beams(orc_ctx* c) {
  Oracle& o = c->o;
  o.items.clear();
  o.BoundPhotonBeams(o.beams_const);     // skeleton.cpp:213-214
  o.BoundPhotonBeams(o.beams_scatt);
  return (int)o.items.size();
}
int orc_build_tree(orc_ctx* c, int max_depth) {
  Oracle& o = c->o;
  o.nodes.clear(); o.segments.clear(); o.segment_leaf.clear(); o.lost_segments.clear();
  o.tree_error = 0; o.max_depth_limit = max_depth > 0 ? max_depth : 4096;
  int root = o.newNode(o.root_aabb, 0);                 // skeleton.cpp:217
  std::vector<PhotonSeg> child = o.items;
  o.BuildTree(root, child, 0);                          // skeleton.cpp:219
  if (o.tree_error) return -1;
  return (int)o.segments.size();
}
int orc_tree_stats(orc_ctx* c, int* n_nodes, int* n_leaves, int* max_depth) {
  int leaves = 0, md = 0;
  for (const Node& n : c->o.nodes) { if (n.leaf) leaves++; md = std::max(md, n.depth); }
  *n_nodes = (int)c->o.nodes.size(); *n_leaves = leaves; *max_depth = md;
  return 0;
}
int orc_lost_segments(orc_ctx* c) { return (int)c->o.lost_segments.size(); }   // CastPhotons.h:84, :1065-1077
void orc_root_box(orc_ctx* c, float lo[3], float hi[3]) {
  const AABB& b = c->o.root_aabb;
  lo[0] = b.min.x; lo[1] = b.min.y; lo[2] = b.max.z;
  hi[0] = b.max.x; hi[1] = b.max.y; hi[2] = b.min.z;
}

static dvec3 lookup_energy(const Oracle& o, const PhotonSeg& s) {
  // BeamRadiance.h:449-455: seg.scattered is never true (CastPhotons.h:398), so
  // the lookup always lands in beams_const, also for segments of beams_scatt.
  const std::vector<PhotonBeam>& list = s.scattered ? o.beams_scatt : o.beams_const;
  if (s.id >= 0 && s.id < (int)list.size()) return list[s.id].energy;
  return dvec3{0, 0, 0};   // the reference would read past the vector here
}

int orc_export_segments(orc_ctx* c, int which, pb_segment* out, int capacity) {
  Oracle& o = c->o;
  const std::vector<PhotonSeg>& src = which ? o.segments : o.items;
  if ((int)src.size() > capacity) return -1;
  for (size_t i = 0; i < src.size(); ++i) {
    const PhotonSeg& s = src[i];
    pb_segment& d = out[i];
    std::memset(&d, 0, sizeof(d));
    d.start[0] = s.start.x; d.start[1] = s.start.y; d.start[2] = s.start.z;
    d.end[0] = s.end.x; d.end[1] = s.end.y; d.end[2] = s.end.z;
    d.orig_start[0] = s.orig_start.x; d.orig_start[1] = s.orig_start.y; d.orig_start[2] = s.orig_start.z;
    d.radius = s.radius;
    vec4 bmin = s.min, bmax = s.max;
    if (which) { bmin = o.nodes[o.segment_leaf[i]].aabb.min; bmax = o.nodes[o.segment_leaf[i]].aabb.max; }
    d.cull_lo[0] = bmin.x; d.cull_lo[1] = bmin.y; d.cull_lo[2] = bmax.z;
    d.cull_hi[0] = bmax.x; d.cull_hi[1] = bmax.y; d.cull_hi[2] = bmin.z;
    dvec3 e = lookup_energy(o, s);
    d.energy[0] = (float)e.x; d.energy[1] = (float)e.y; d.energy[2] = (float)e.z;
    d.s_ext = s.s_ext; d.e_ext = s.e_ext; d.c_ext = s.c_ext;
    d.seg_id = s.seg_id; d.id = s.id;
    d.flags = (s.ada_width ? 1u : 0u) | (s.scattered ? 2u : 0u);
  }
  return (int)src.size();
}

int orc_set_grid(orc_ctx* c, const float* sigma, int side) {
  c->o.GRID = grid_from_sigma(sigma, side);
  return 0;
}
// HeterogeneousRun, skeleton.cpp:113-125: CreateSurface then GRID <- matrix * GRID.
int orc_make_grid(orc_ctx* c, int triangle_number, float height, double time, float* sigma_out) {
  Oracle& o = c->o;
  CreateSurface(o.GRID, triangle_number, height, time);
  for (vec4& p : o.GRID.geometric_points) p = o.matrix * p;
  if (sigma_out) for (size_t i = 0; i < o.GRID.geometric_points.size(); ++i) sigma_out[i] = o.GRID.geometric_points[i].z;
  return o.GRID.side_points;
}

// BeamRadianceStandard / BVHBeamRadianceHet as the reference recurses them.
static void traverse(const Oracle& o, int node, const RayCtx& r, int mode, const Medium& med, dvec3& current,
                     orc_counters& cnt) {
  cnt.node_visits++;
  const Node& n = o.nodes[node];
  if (!box_gate(n.aabb, r)) return;
  if (n.leaf) {
    for (int i = 0; i < 2; ++i) {
      if (n.seg[i] < 0) continue;
      const PhotonSeg& seg = o.segments[n.seg[i]];
      leaf_segment(seg, lookup_energy(o, seg), r, mode, med, current, cnt);
    }
  }
  if (n.left >= 0) traverse(o, n.left, r, mode, med, current, cnt);
  if (n.right >= 0) traverse(o, n.right, r, mode, med, current, cnt);
}

static void add_counters(orc_counters& a, const orc_counters& b) {
  a.rays += b.rays; a.rays_hit_scene += b.rays_hit_scene; a.node_visits += b.node_visits;
  a.cyl_calls += b.cyl_calls; a.cyl_true += b.cyl_true; a.valid_hits += b.valid_hits;
}

// ---- CPU twin of the GPU's linear BVH (keys, stable sort, Karras hierarchy, refit) ----
static uint64_t expand16(uint32_t v) {
  uint64_t x = v & 0xFFFFu;
  x = (x | (x << 16)) & 0x000000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
static int lbvh_delta(const std::vector<uint64_t>& keys, int i, int j) {
  const int n = (int)keys.size();
  if (j < 0 || j >= n) return -1;
  const uint64_t x = keys[i] ^ keys[j];
  if (x == 0ull) return 64 + __builtin_clz((uint32_t)(i ^ j));
  return __builtin_clzll(x);
}
int orc_lbvh(const pb_segment* segs, int n, uint64_t* keys_out, uint32_t* perm_out, pb_node* nodes_out) {
  if (n < 2) return 0;
  std::vector<float> mid(3 * (size_t)n);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      const float m = (segs[i].start[k] + segs[i].end[k]) * 0.5f;
      mid[3 * (size_t)i + k] = m;
      lo[k] = std::fmin(lo[k], m); hi[k] = std::fmax(hi[k], m);
    }
  std::vector<uint64_t> key(n);
  for (int i = 0; i < n; ++i) {
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
      const float ext = hi[k] - lo[k];
      const float scale = (ext > 0.0f) ? (65535.0f / ext) : 0.0f;
      float t = (mid[3 * (size_t)i + k] - lo[k]) * scale;
      t = std::fmin(std::fmax(t, 0.0f), 65535.0f);
      q[k] = (uint32_t)t;
    }
    key[i] = (expand16(q[0]) << 2) | (expand16(q[1]) << 1) | expand16(q[2]);
  }
  std::vector<uint32_t> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = (uint32_t)i;
  std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  std::vector<uint64_t> keys(n);
  for (int i = 0; i < n; ++i) keys[i] = key[perm[i]];
  std::vector<int> left(n - 1), right(n - 1), parent(n - 1, -1), parent_leaf(n, -1);
  for (int i = 0; i < n - 1; ++i) {
    const int d = (lbvh_delta(keys, i, i + 1) - lbvh_delta(keys, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
      if (lbvh_delta(keys, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
      if (lbvh_delta(keys, i, i + (s + t) * d) > dnode) s += t;
      if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int a = std::min(i, j), b = std::max(i, j);
    left[i] = (a == gamma) ? ~gamma : gamma;
    right[i] = (b == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    if (left[i] >= 0) parent[left[i]] = i; else parent_leaf[~left[i]] = i;
    if (right[i] >= 0) parent[right[i]] = i; else parent_leaf[~right[i]] = i;
  }
  // boxes: leaves are padded capsule bounds, inner nodes the union of their children (post-order)
  std::vector<float> leaf_box(6 * (size_t)n), node_box(6 * ((size_t)n - 1));
  for (int i = 0; i < n; ++i) {
    const pb_segment& sg = segs[perm[i]];
    const float pad = sg.radius * 1.001f + 1e-5f;
    for (int k = 0; k < 3; ++k) {
      // (fixed-radius segments: an adaptive-width one is bounded by a capsule around its gate box instead)
      leaf_box[6 * (size_t)i + k] = std::fmax(std::fmin(std::fmin(sg.start[k], sg.end[k]) - pad, 1e30f), -1e30f);
      leaf_box[6 * (size_t)i + 3 + k] = std::fmax(std::fmin(std::fmax(sg.start[k], sg.end[k]) + pad, 1e30f), -1e30f);
    }
  }
  std::vector<int> order; order.reserve(n - 1);
  { std::vector<int> st; st.push_back(0);
    while (!st.empty()) { int v = st.back(); st.pop_back(); order.push_back(v);
      if (left[v] >= 0) st.push_back(left[v]);
      if (right[v] >= 0) st.push_back(right[v]); } }
  for (int q = (int)order.size() - 1; q >= 0; --q) {
    const int v = order[q];
    const float* a = left[v] >= 0 ? &node_box[6 * (size_t)left[v]] : &leaf_box[6 * (size_t)(~left[v])];
    const float* b = right[v] >= 0 ? &node_box[6 * (size_t)right[v]] : &leaf_box[6 * (size_t)(~right[v])];
    for (int k = 0; k < 3; ++k) {
      node_box[6 * (size_t)v + k] = std::fmin(a[k], b[k]);
      node_box[6 * (size_t)v + 3 + k] = std::fmax(a[3 + k], b[3 + k]);
    }
  }
  if ((int)order.size() != n - 1) return -1;   // the hierarchy must reach every inner node exactly once
  for (int i = 0; i < n; ++i) { if (keys_out) keys_out[i] = keys[i]; if (perm_out) perm_out[i] = perm[i]; }
  if (nodes_out)
    for (int i = 0; i < n - 1; ++i) {
      for (int k = 0; k < 3; ++k) { nodes_out[i].lo[k] = node_box[6 * (size_t)i + k]; nodes_out[i].hi[k] = node_box[6 * (size_t)i + 3 + k]; }
      nodes_out[i].left = left[i]; nodes_out[i].right = right[i];
    }
  return n;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int orc_render_tree(orc_ctx* c, int x0, int y0, int w, int h, int mode, double* rgb, orc_counters* counters,
                    int threads) {
  const Oracle& o = c->o;
  if (o.nodes.empty()) return -1;
  const int ssaa = o.ctl.c.ssaa;
  Medium med{o.extinction_c(), o.scattering_c(), &o.GRID, o.ctl.c.one_dimensional != 0};
  orc_counters total{};
  (void)threads;
  ViewDraws view;
  view.kind = o.rng.kind; view.seed = o.ctl.c.seed; view.sub_h = o.ctl.c.screen_height * ssaa;
  if (o.ctl.c.short_view && view.kind == 0) {
    // the generator continues where the photon pass and BuildTree left it; one draw per camera ray that
    // hits the scene, in Draw's loop order (skeleton.cpp:421-443)
    std::mt19937 generator = o.rng.generator;
    std::uniform_real_distribution<double> uniform(0.0, 1.0);
    const int sub_w = o.ctl.c.screen_width * ssaa;
    view.table.assign((size_t)sub_w * view.sub_h, std::nan(""));
    for (int x = 0; x < sub_w; x += ssaa)
      for (int y = 0; y < view.sub_h; y += ssaa)
        for (int i = 0; i < ssaa; ++i)
          for (int j = 0; j < ssaa; ++j) {
            RayCtx r;
            if (camera_ray(o.ctl, o.triangles, o.matrix, x + i, y + j, r))
              view.table[(size_t)(x + i) * view.sub_h + (y + j)] = uniform(generator);
          }
  }
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_counters cnt{};
#pragma omp for schedule(dynamic, 1)
    for (int yy = 0; yy < h; ++yy)
      for (int xx = 0; xx < w; ++xx) {
        dvec3 current{0, 0, 0};
        for (int i = 0; i < ssaa; ++i)
          for (int j = 0; j < ssaa; ++j) {
            RayCtx r;
            cnt.rays++;
            if (camera_ray(o.ctl, o.triangles, o.matrix, (x0 + xx) * ssaa + i, (y0 + yy) * ssaa + j, r, &view)) {
              cnt.rays_hit_scene++;
              traverse(o, 0, r, mode, med, current, cnt);
            }
          }
        double* p = rgb + ((size_t)yy * w + xx) * 3;
        p[0] = current.x / (double)ssaa; p[1] = current.y / (double)ssaa; p[2] = current.z / (double)ssaa;
      }
#pragma omp critical
    add_counters(total, cnt);
  }
  if (counters) *counters = total;
  return 0;
}

int orc_render_segments_strided(const pb_control* control, const float* tri_xyz, int n_tri, const float m16[16],
                                const float* grid_sigma, int side, const pb_segment* segs, int n,
                                const float root_lo[3], const float root_hi[3], int x0, int y0, int w, int h,
                                int step_x, int step_y, int mode, double* rgb, orc_counters* counters, int threads);

int orc_render_segments(const pb_control* control, const float* tri_xyz, int n_tri, const float m16[16],
                        const float* grid_sigma, int side, const pb_segment* segs, int n, const float root_lo[3],
                        const float root_hi[3], int x0, int y0, int w, int h, int mode, double* rgb,
                        orc_counters* counters, int threads) {
  return orc_render_segments_strided(control, tri_xyz, n_tri, m16, grid_sigma, side, segs, n, root_lo, root_hi, x0, y0,
                                     w, h, 1, 1, mode, rgb, counters, threads);
}

// Same estimate on the pixel lattice (x0 + i*step_x, y0 + j*step_y), i < w, j < h.
int orc_render_segments_strided(const pb_control* control, const float* tri_xyz, int n_tri, const float m16[16],
                                const float* grid_sigma, int side, const pb_segment* segs, int n,
                                const float root_lo[3], const float root_hi[3], int x0, int y0, int w, int h,
                                int step_x, int step_y, int mode, double* rgb, orc_counters* counters, int threads) {
  Control ctl; ctl.c = *control;
  std::vector<Triangle> tris = tris_from_xyz(tri_xyz, n_tri);
  mat4 M; std::memcpy(M.m, m16, sizeof(float) * 16);
  Grid grid;
  if (grid_sigma && side > 1) grid = grid_from_sigma(grid_sigma, side);
  if (mode == PB_MODE_73HET && grid.side_points < 2) return -1;
  Medium med{ctl.extinction_c(), ctl.c.scattering_c, &grid, ctl.c.one_dimensional != 0};
  std::vector<PhotonSeg> ps(n);
  std::vector<AABB> boxes(n);
  std::vector<dvec3> energy(n);
  for (int i = 0; i < n; ++i) {
    ps[i] = seg_from_record(segs[i]);
    boxes[i] = box_from_lohi(segs[i].cull_lo, segs[i].cull_hi);
    energy[i] = dvec3{(double)segs[i].energy[0], (double)segs[i].energy[1], (double)segs[i].energy[2]};
  }
  AABB root = box_from_lohi(root_lo, root_hi);
  const int ssaa = ctl.c.ssaa;
  orc_counters total{};
  (void)threads;
  ViewDraws view;                     // no generator state here: the device's per-ray draws
  view.kind = 1; view.seed = ctl.c.seed; view.sub_h = ctl.c.screen_height * ssaa;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_counters cnt{};
#pragma omp for schedule(dynamic, 1)
    for (int yy = 0; yy < h; ++yy)
      for (int xx = 0; xx < w; ++xx) {
        dvec3 current{0, 0, 0};
        for (int i = 0; i < ssaa; ++i)
          for (int j = 0; j < ssaa; ++j) {
            RayCtx r;
            cnt.rays++;
            if (!camera_ray(ctl, tris, M, (x0 + xx * step_x) * ssaa + i, (y0 + yy * step_y) * ssaa + j, r, &view)) continue;
            cnt.rays_hit_scene++;
            cnt.node_visits++;
            if (!box_gate(root, r)) continue;
            for (int s = 0; s < n; ++s) {
              cnt.node_visits++;
              if (!box_gate(boxes[s], r)) continue;
              leaf_segment(ps[s], energy[s], r, mode, med, current, cnt);
            }
          }
        double* p = rgb + ((size_t)yy * w + xx) * 3;
        p[0] = current.x / (double)ssaa; p[1] = current.y / (double)ssaa; p[2] = current.z / (double)ssaa;
      }
#pragma omp critical
    add_counters(total, cnt);
  }
  if (counters) *counters = total;
  return 0;
}

}  // extern "C"
