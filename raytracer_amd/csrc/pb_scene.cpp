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
  if (!tri_xyz) return PB_ERR_ARG;
  if (capacity < 30) return PB_ERR_CAPACITY;
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

// ----------------------------------------------------------------------------------------------
// Extinction field of the heterogeneous medium.
//
// CreateSurface + UpdateHeight (Tessendorf.h:47-132) lay a (triangle_number+1)^2 lattice over
// [-1,1]^2 and set z = |sum over a 5x5 block of wave vectors of h(k,t) * exp(i k.x)| + height.  The
// spectrum term h(k,t) = sqrt(P(k)/2) xi e^{i w t} + sqrt(P(-k)/2) xi e^{-i w t} does not depend on
// the lattice point, so it is tabulated once per wave vector; the per-point sum then runs over the
// table in the reference's (n, m) loop order with the same float operations.
// ----------------------------------------------------------------------------------------------
#include <complex>
#include <vector>

extern "C" int pb_create_surface(int triangle_number, float height, double time, float* points_xyzw, int capacity_points) {
  if (triangle_number < 1 || !points_xyzw) return PB_ERR_ARG;
  typedef std::complex<float> cf;
  const float width = 2.0f, half = width / 2;                        // ACTUAL_WIDTH, HALF_W  :23-24
  const float gravity = 9.8f, max_wave = 0.4f, amplitude = 0.2f;     // :31-34
  const float wind[2] = {0.424f, 0.316f};
  const float dk = half / 5;                                         // `step`  :36
  const double xi_re = 0.5, xi_im = 0.5;                             // e_r, e_i  :40-41
  const double pi = 3.14159265;                                      // :4
  const int side = triangle_number + 1;
  const int total = side * side;
  if (capacity_points < total) return PB_ERR_CAPACITY;

  struct Wave { float kx, kz; cf h; };
  std::vector<Wave> table;
  const float inv_sqrt2 = (float)(1 / std::sqrt(2.0));               // `multiplier`  :83
  for (float n = 0; n < half; n = n + dk) {                          // float counters decide the wave count
    for (float m = 0; m < half; m = m + dk) {
      Wave wv;
      wv.kx = (float)((2 * pi * (n - half / 2)) / max_wave);
      wv.kz = (float)((2 * pi * (m - half / 2)) / max_wave);
      const float k2 = wv.kx * wv.kx + wv.kz * wv.kz;
      const float k_len = std::sqrt(k2);
      const float omega = std::sqrt(gravity * k_len);
      const float k_len4 = (float)std::pow((double)k_len, 4);
      const cf backward((float)std::cos(-time * omega), (float)std::sin(-time * omega));
      const cf forward((float)std::cos(time * omega), (float)std::sin(time * omega));
      float p_plus = 0, p_minus = 0;                                 // Phillips spectrum for k and -k
      if (k_len4 != 0) {
        const float along = wv.kx * wind[0] + wv.kz * wind[1];
        const float against = (-wv.kx) * wind[0] + (-wv.kz) * wind[1];
        p_plus = (float)(amplitude * (std::exp(-1 / k2) / k_len4) * std::pow((double)std::fabs(along), 2));
        p_minus = (float)(amplitude * (std::exp(-1 / k2) / k_len4) * std::pow((double)std::fabs(against), 2));
      }
      const cf xi((float)xi_re, (float)xi_im);
      const cf a = inv_sqrt2 * xi * cf(std::sqrt(p_plus));
      const cf b = inv_sqrt2 * xi * cf(std::sqrt(p_minus));
      wv.h = (a * forward) + (b * backward);
      table.push_back(wv);
    }
  }
  const float pitch = width / (float)triangle_number;
  for (int i = 0; i < total; ++i) {
    // lattice point, shifted by (HALF_W, HALF_W)  :57-66
    const float x = (i % side) * pitch - half;
    const float y = (float)std::floor((double)(i / side)) * pitch - half;
    cf sum = 0;
    for (const Wave& wv : table) {
      const float phase = wv.kx * x + wv.kz * y;
      sum += wv.h * cf(std::cos(phase), std::sin(phase));
    }
    float z = std::abs(sum);
    z += height;
    points_xyzw[i * 4 + 0] = x; points_xyzw[i * 4 + 1] = y; points_xyzw[i * 4 + 2] = z; points_xyzw[i * 4 + 3] = 1.0f;
  }
  return total;
}

extern "C" void pb_transform_points(const float M[16], float* p, int n) {
  for (int i = 0; i < n; ++i) {
    const float x = p[i * 4], y = p[i * 4 + 1], z = p[i * 4 + 2], w = p[i * 4 + 3];
    // glm mat4 * vec4: (col0*x + col1*y) + (col2*z + col3*w)   type_mat4x4.inl:641-652
    for (int r = 0; r < 4; ++r) p[i * 4 + r] = (M[r] * x + M[4 + r] * y) + (M[8 + r] * z + M[12 + r] * w);
  }
}
