// Device geometry shared by the render and emission kernels.
#ifndef PB_GEOM_H
#define PB_GEOM_H
#include "pb_internal.h"

// ClosestIntersection, Intersections.h:458-497, on prepared triangles.
// screen_box (optional, rays from the camera at (0,0,0) only): per triangle the bounds of x/z and y/z over
// its vertices, widened by 1e-3.  A ray outside them cannot satisfy the barycentric tests (they hold to a
// few 1e-6), so the triangle is skipped without evaluating it; the result is unchanged.
template <class TriPtr, class BoxPtr>
PB_DI bool closest_intersection(TriPtr tris, int n_tri, V4 start, V4 dir, V4& position,
                                float& distance, int& index, BoxPtr screen_box, unsigned long long only = ~0ull) {
  bool found = false;
  float best = 3.402823466e+38f;   // Intersections.h:57
  int best_i = -1;
  const float ndx = -dir.x, ndy = -dir.y, ndz = -dir.z;
  auto test = [&](int i) {
    if (screen_box) {
      const float sb_x = screen_box[i].x, sb_y = screen_box[i].y, sb_z = screen_box[i].z, sb_w = screen_box[i].w;
      if (dir.x < sb_x * dir.z || dir.x > sb_y * dir.z || dir.y < sb_z * dir.z || dir.y > sb_w * dir.z) return;
    }
    const auto& T = tris[i];
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
  };
  // `only` (same for the whole wave): triangles 0..63 the caller has not ruled out for all of its rays; visited in
  // ascending order like the reference's loop (the first of two equal distances wins)
  unsigned long long todo = only;
  if (n_tri < 64) todo &= (1ull << n_tri) - 1ull;
  while (todo) {
    const int i = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    test(i);
  }
  for (int i = 64; i < n_tri; ++i) test(i);
  distance = best;
  index = best_i;
  if (found) position = start + dir * best;
  return found;
}

template <class TriPtr>
PB_DI bool closest_intersection(TriPtr tris, int n_tri, V4 start, V4 dir, V4& position, float& distance, int& index) {
  return closest_intersection(tris, n_tri, start, dir, position, distance, index, (const float4*)nullptr);
}

#endif
