// TEST INFRASTRUCTURE — thin C wrappers around the REFERENCE's own functions.
// Builds only where /root/reference exists (see oracle/Makefile, target _ref);
// it #includes the reference headers where they lie and contains no reference
// code.  Only the SDL-free headers are used: Control.h, TestModelH.h,
// Tessendorf.h, Intersections.h and Analysis.h.  CastPhotons.h, BeamRadiance.h and
// Heterogeneous.h include <SDL.h> (absent in this image) and are not built.
#include <glm/glm.hpp>
#include <cstring>
#include <iostream>
#include <limits>
#include <random>
#include <vector>

#include "Control.h"
#include "TestModelH.h"
#include "Tessendorf.h"
#include "Intersections.h"
#include "Analysis.h"

static vec4 v4(const float* p) { return vec4(p[0], p[1], p[2], p[3]); }

extern "C" {

int ref_load_test_model(float* tri, float* normals, int capacity) {
  std::vector<Triangle> t;
  LoadTestModel(t);
  if ((int)t.size() > capacity) return -1;
  for (size_t i = 0; i < t.size(); ++i) {
    const vec4* vs[3] = {&t[i].v0, &t[i].v1, &t[i].v2};
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 4; ++c) tri[i * 12 + k * 4 + c] = (*vs[k])[c];
    if (normals)
      for (int c = 0; c < 4; ++c) normals[i * 4 + c] = t[i].normal[c];
  }
  return (int)t.size();
}

int ref_closest_intersection(const float start[4], const float dir[4], const float m16[16], const float* tri,
                             int n_tri, float position[4], float* distance, int* index) {
  mat4 M;
  std::memcpy(&M[0][0], m16, sizeof(float) * 16);
  std::vector<Triangle> t;
  for (int i = 0; i < n_tri; ++i)
    t.push_back(Triangle(v4(tri + i * 12), v4(tri + i * 12 + 4), v4(tri + i * 12 + 8), vec3(1, 1, 1), false, false));
  Intersection ci;
  ci.position = vec4(0, 0, 0, 0);
  ci.index = -1;
  bool f = ClosestIntersection(v4(start), v4(dir), ci, M, t);
  for (int c = 0; c < 4; ++c) position[c] = ci.position[c];
  *distance = ci.distance;
  *index = ci.index;
  return f ? 1 : 0;
}

int ref_hit_bounding_box(const float bmin[4], const float bmax[4], const float start[4], const float dir[4],
                         float hit[4]) {
  AABB b;
  b.min = v4(bmin);
  b.max = v4(bmax);
  b.mid = (b.min + b.max) / 2.0f;
  vec4 h(0, 0, 0, 0);
  bool r = HitBoundingBox(b, v4(start), v4(dir), h);
  for (int c = 0; c < 4; ++c) hit[c] = h[c];
  return r ? 1 : 0;
}

static void hit_quadric(bool cone, const float start[4], const float dir[4], const float s0[4], const float s1[4],
                        const float so[4], float radius, float out[12]) {
  PhotonSeg seg;
  seg.start = v4(s0);
  seg.end = v4(s1);
  seg.orig_start = v4(so);
  seg.radius = radius;
  seg.ada_width = cone;
  CylIntersection ci;
  ci.valid = false;
  bool r = cone ? HitCone(v4(start), v4(dir), seg, ci) : HitCylinder(v4(start), v4(dir), seg, ci);
  for (int i = 0; i < 12; ++i) out[i] = 0;
  out[0] = r ? 1.f : 0.f;
  if (r) {
    out[1] = ci.valid ? 1.f : 0.f;
    if (ci.valid) {
      out[2] = ci.tb_minus; out[3] = ci.tb_plus; out[4] = ci.tc_minus; out[5] = ci.tc_plus;
      for (int c = 0; c < 3; ++c) { out[6 + c] = ci.entry_point[c]; out[9 + c] = ci.exit_point[c]; }
    }
  }
}

void ref_hit_cylinder(const float start[4], const float dir[4], const float s0[4], const float s1[4],
                      const float so[4], float radius, float out[12]) {
  hit_quadric(false, start, dir, s0, s1, so, radius, out);
}

void ref_hit_cone(const float start[4], const float dir[4], const float s0[4], const float s1[4],
                  const float so[4], float radius, float out[12]) {
  hit_quadric(true, start, dir, s0, s1, so, radius, out);
}

void ref_find_rotation_matrix(const float c[3], const float w[3], float m9[9]) {
  mat3 R = findRotationMatrix(vec3(c[0], c[1], c[2]), vec3(w[0], w[1], w[2]));
  std::memcpy(m9, &R[0][0], sizeof(float) * 9);
}

int ref_create_surface(int triangle_number, float height, double time, float* pts) {
  CreateSurface(triangle_number, height, time);
  for (size_t i = 0; i < GRID.geometric_points.size(); ++i)
    for (int c = 0; c < 4; ++c) pts[i * 4 + c] = GRID.geometric_points[i][c];
  return (int)GRID.geometric_points.size();
}

// RunAnalysis (Analysis.h:81-122) writes NSD.csv and rRMSE.csv into the current directory.
void ref_run_analysis(void) { RunAnalysis(); }

void ref_control(float* out /*12*/) {
  for (int c = 0; c < 4; ++c) out[c] = light_position[c];
  for (int c = 0; c < 3; ++c) out[4 + c] = light_power[c];
  out[7] = absorption_c; out[8] = scattering_c; out[9] = extinction_c;
  out[10] = (float)PI; out[11] = ACTUAL_WIDTH;
}

}  // extern "C"
