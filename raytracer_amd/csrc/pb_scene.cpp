// Host-side scene set-up: the Cornell box of LoadTestModel (TestModelH.h:45-214)
// as a vertex/index table.  One-time setup, as in the reference's main().
#include "../../include/photon_beams.h"

namespace {

struct P3 { float x, y, z; };

// corner order: A B C D (floor y = 0), E F G H (top y = h), same x/z
void box_corners(const float xz[4][2], float h, P3 c[8]) {
  for (int i = 0; i < 4; ++i) {
    c[i] = P3{xz[i][0], 0.0f, xz[i][1]};
    c[i + 4] = P3{xz[i][0], h, xz[i][1]};
  }
}

enum { A, B, C, D, E, F, G, H };

// TestModelH.h:101-118 (room; no front wall)
const int kRoom[10][3] = {{E, F, G}, {F, H, G}, {C, B, A}, {C, D, B}, {A, E, C},
                          {C, E, G}, {F, B, D}, {H, F, D}, {G, D, C}, {G, H, D}};
// TestModelH.h:134-151 and :167-184 (the two blocks share one pattern)
const int kBlock[10][3] = {{E, B, A}, {E, F, B}, {F, D, B}, {F, H, D}, {H, C, D},
                           {H, G, C}, {G, E, C}, {E, A, C}, {G, F, E}, {G, H, F}};

}  // namespace

extern "C" int pb_load_test_model(float* tri_xyz, float* normals_xyz, int capacity) {
  if (!tri_xyz || capacity < 30) return PB_ERR_ARG;
  const float L = 555;                                   // TestModelH.h:65
  const float room_xz[4][2] = {{L, 0}, {0, 0}, {L, L}, {0, L}};
  const float short_xz[4][2] = {{290, 114}, {130, 65}, {240, 272}, {82, 225}};   // :123-126
  const float tall_xz[4][2] = {{423, 247}, {265, 296}, {472, 406}, {314, 456}};  // :156-159
  P3 corners[3][8];
  box_corners(room_xz, L, corners[0]);
  box_corners(short_xz, 165, corners[1]);
  box_corners(tall_xz, 330, corners[2]);
  const float s = 2 / L;                                 // :192
  int n = 0;
  for (int obj = 0; obj < 3; ++obj) {
    const int(*idx)[3] = (obj == 0) ? kRoom : kBlock;
    for (int t = 0; t < 10; ++t, ++n) {
      float v[3][3];
      for (int k = 0; k < 3; ++k) {
        const P3& c = corners[obj][idx[t][k]];
        // scale to [-1,1]^3, then flip x and y  (:190-206)
        v[k][0] = -((c.x * s) - 1);
        v[k][1] = -((c.y * s) - 1);
        v[k][2] = (c.z * s) - 1;
        for (int a = 0; a < 3; ++a) tri_xyz[n * 9 + k * 3 + a] = v[k][a];
      }
      if (normals_xyz) {
        // Triangle::ComputeNormal, :29-38: normalize(cross(e2, e1))
        float e1[3], e2[3];
        for (int a = 0; a < 3; ++a) { e1[a] = v[1][a] - v[0][a]; e2[a] = v[2][a] - v[0][a]; }
        float cx = e2[1] * e1[2] - e1[1] * e2[2];
        float cy = e2[2] * e1[0] - e1[2] * e2[0];
        float cz = e2[0] * e1[1] - e1[0] * e2[1];
        float inv = 1.0f / __builtin_sqrtf(cx * cx + cy * cy + cz * cz);
        normals_xyz[n * 3 + 0] = cx * inv; normals_xyz[n * 3 + 1] = cy * inv; normals_xyz[n * 3 + 2] = cz * inv;
      }
    }
  }
  return n;
}
