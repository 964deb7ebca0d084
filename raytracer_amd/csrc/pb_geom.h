// Device geometry shared by the render and emission kernels.
#ifndef PB_GEOM_H
#define PB_GEOM_H
#include "pb_internal.h"

// ClosestIntersection, Intersections.h:458-497, on prepared triangles.
// screen_box (optional, rays from the camera at (0,0,0) only): per triangle the bounds of x/z and y/z over
// its vertices, widened by 1e-3.  A ray outside them cannot satisfy the barycentric tests (they hold to a
// few 1e-6), so the triangle is skipped without evaluating it; the result is unchanged.
PB_DI bool closest_intersection(const TriPrep* __restrict__ tris, int n_tri, V4 start, V4 dir, V4& position,
                                float& distance, int& index, const float4* __restrict__ screen_box = nullptr,
                                unsigned long long only = ~0ull) {
  bool found = false;
  float best = 3.402823466e+38f;   // Intersections.h:57
  int best_i = -1;
  const float ndx = -dir.x, ndy = -dir.y, ndz = -dir.z;
  for (int i = 0; i < n_tri; ++i) {
    // `only` (same for the whole wave): triangles 0..63 the caller has ruled out for all of its rays
    if (i < 64 && !((only >> i) & 1ull)) continue;
    if (screen_box) {
      const float4 sb = screen_box[i];
      if (dir.x < sb.x * dir.z || dir.x > sb.y * dir.z || dir.y < sb.z * dir.z || dir.y > sb.w * dir.z) continue;
    }
    const TriPrep& T = tris[i];
    const float bx = start.x - T.v0[0], by = start.y - T.v0[1], bz = start.z - T.v0[2];
    const float e1x = T.e1[0], e1y = T.e1[1], e1z = T.e1[2];
    const float e2x = T.e2[0], e2y = T.e2[1], e2z = T.e2[2];
    const float P = ndy * e2z - e2y * ndz;
    const float Q = ndy * e1z - e1y * ndz;
    const float det = ndx * T.c0 - e1x * P + e2x * Q;
    const float ood = 1.0f / det;
    const float i00 = T.c0 * ood;
    const float i10 = -T.c1 * ood;
    const float i20 = T.c2 * ood;
    const float i01 = -P * ood;
    const float i11 = (ndx * e2z - e2x * ndz) * ood;
    const float i21 = -(ndx * e2y - e2x * ndy) * ood;
    const float i02 = Q * ood;
    const float i12 = -(ndx * e1z - e1x * ndz) * ood;
    const float i22 = (ndx * e1y - e1x * ndy) * ood;
    const float xx = i00 * bx + i10 * by + i20 * bz;
    const float xy = i01 * bx + i11 * by + i21 * bz;
    const float xz = i02 * bx + i12 * by + i22 * bz;
    if (xx >= 0 && xy >= 0 && xz >= 0 && (xy + xz) < 1) {
      if (xx < best) { found = true; best = xx; best_i = i; }
    }
  }
  distance = best;
  index = best_i;
  if (found) position = start + dir * best;
  return found;
}

#endif
