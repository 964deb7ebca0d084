// Per-scene and per-segment precomputation.  Everything here is a function of
// the scene or of one segment only; the reference recomputes it for every
// (ray, triangle) and (ray, segment) pair.  The arithmetic keeps the reference's
// operation order (file built with -ffp-contract=off), so the per-pair kernels
// see bit-identical operands.
#include "pb_internal.h"

// ClosestIntersection, Intersections.h:468-476: v = matrix * triangle vertex,
// e1 = v1 - v0, e2 = v2 - v0, and the cofactors of columns 1,2 of A = (-dir,e1,e2).
__global__ void prepare_scene_kernel(const float* __restrict__ tri_xyz, int n_tri, const float* __restrict__ matrix,
                                     TriPrep* __restrict__ out, float4* __restrict__ screen_box, float* __restrict__ screen_edge) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tri) return;
  float m[16];
  for (int k = 0; k < 16; ++k) m[k] = matrix[k];
  const float* t = tri_xyz + (size_t)i * 9;
  V4 w0 = v4(t[0], t[1], t[2], 1.0f), w1 = v4(t[3], t[4], t[5], 1.0f), w2 = v4(t[6], t[7], t[8], 1.0f);
  V4 v0 = mul4(m, w0), v1 = mul4(m, w1), v2 = mul4(m, w2);
  V3 e1 = v3(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z);
  V3 e2 = v3(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z);
  TriPrep p;
  p.v0[0] = v0.x; p.v0[1] = v0.y; p.v0[2] = v0.z;
  p.e1[0] = e1.x; p.e1[1] = e1.y; p.e1[2] = e1.z;
  p.e2[0] = e2.x; p.e2[1] = e2.y; p.e2[2] = e2.z;
  p.c0 = e1.y * e2.z - e2.y * e1.z;
  p.c1 = e1.x * e2.z - e2.x * e1.z;
  p.c2 = e1.x * e2.y - e2.x * e1.y;
  // Triangle::ComputeNormal, TestModelH.h:29-38 (world-space vertices)
  V3 f1 = v3(w1.x - w0.x, w1.y - w0.y, w1.z - w0.z);
  V3 f2 = v3(w2.x - w0.x, w2.y - w0.y, w2.z - w0.z);
  V3 n = normalize3(cross3(f2, f1));
  p.n[0] = n.x; p.n[1] = n.y; p.n[2] = n.z;
  p.pad = 0.0f;
  out[i] = p;
  // bounds of x/z and y/z seen from the camera at (0,0,0); unbounded if a vertex is beside or behind it
  float4 sb = make_float4(-3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, 3.402823466e+38f);
  if (v0.z > 0.05f && v1.z > 0.05f && v2.z > 0.05f) {
    const float u0 = v0.x / v0.z, u1 = v1.x / v1.z, u2 = v2.x / v2.z;
    const float w0 = v0.y / v0.z, w1 = v1.y / v1.z, w2 = v2.y / v2.z;
    const float m = 1e-3f;
    sb = make_float4(fminf(u0, fminf(u1, u2)) - m, fmaxf(u0, fmaxf(u1, u2)) + m,
                     fminf(w0, fminf(w1, w2)) - m, fmaxf(w0, fmaxf(w1, w2)) + m);
  }
  screen_box[i] = sb;
  // The triangle's three edges as lines in the same (x/z, y/z) plane (a perspective view keeps lines straight):
  // a u + b v + c = distance from the edge, positive on the side of the third vertex.  A tile whose rectangle lies
  // more than a margin outside one edge cannot hold a ray that satisfies the barycentric tests.  [9] = 1 if usable.
  float e[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (v0.z > 0.05f && v1.z > 0.05f && v2.z > 0.05f) {
    const float u[3] = {v0.x / v0.z, v1.x / v1.z, v2.x / v2.z};
    const float w[3] = {v0.y / v0.z, v1.y / v1.z, v2.y / v2.z};
    bool ok = true;
    for (int k = 0; k < 3; ++k) {
      const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      float a = w[k1] - w[k], b = -(u[k1] - u[k]);
      const float len = sqrtf(a * a + b * b);
      if (!(len > 1e-6f)) { ok = false; break; }
      a /= len; b /= len;
      float c = -(a * u[k] + b * w[k]);
      const float third = a * u[k2] + b * w[k2] + c;
      if (fabsf(third) < 1e-5f) { ok = false; break; }        // seen edge-on: no inside to speak of
      if (third < 0) { a = -a; b = -b; c = -c; }
      e[3 * k] = a; e[3 * k + 1] = b; e[3 * k + 2] = c;
    }
    e[9] = ok ? 1.0f : 0.0f;
  }
  for (int k = 0; k < 12; ++k) screen_edge[(size_t)i * 12 + k] = e[k];
}

// findRotationMatrix, Intersections.h:499-513.
__device__ M3 find_rotation_matrix(V3 c_dir, V3 w_dir) {
  V3 v = cross3(c_dir, w_dir);
  float c = dot3(c_dir, w_dir);
  M3 vx;
  vx.m[0][0] = 0;    vx.m[1][0] = -v.z; vx.m[2][0] = v.y;
  vx.m[0][1] = v.z;  vx.m[1][1] = 0;    vx.m[2][1] = -v.x;
  vx.m[0][2] = -v.y; vx.m[1][2] = v.x;  vx.m[2][2] = 0;
  M3 vx2 = mul(vx, vx);
  float k = 1 / (1 + c);
  M3 R;
  for (int col = 0; col < 3; ++col)
    for (int row = 0; row < 3; ++row) {
      float id = (col == row) ? 1.0f : 0.0f;
      R.m[col][row] = (id + vx.m[col][row]) + (vx2.m[col][row] * k);
    }
  return R;
}

// Candidate planes and quadrant flags of HitBoundingBox (Intersections.h:95-112)
// for the fixed ray origin (0,0,0).
__device__ void gate_setup(const float lo[3], const float hi[3], float cand[3], uint32_t& qflags) {
  uint32_t q = 0;
  bool inside = true;
  for (int i = 0; i < 3; ++i) {
    if (0.0f < lo[i]) { cand[i] = lo[i]; inside = false; }
    else if (0.0f > hi[i]) { cand[i] = hi[i]; inside = false; }
    else { cand[i] = 0.0f; q |= (1u << i); }
  }
  if (inside) q |= 8u;
  qflags = q;
}

// Largest distance from a corner of the box to the segment a-b.
__device__ float box_reach(const float lo[3], const float hi[3], V3 a, V3 b) {
  const V3 ab = b - a;
  const float ll = dot3(ab, ab);
  float far = 0.0f;
  for (int k = 0; k < 8; ++k) {
    const V3 c = v3((k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]);
    float t = ll > 0.0f ? dot3(c - a, ab) / ll : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    far = fmaxf(far, length3(c - (a + ab * t)));
  }
  return far;
}

__global__ void prepare_segments_kernel(const pb_segment* __restrict__ segs, int n, int mode, int one_dimensional, float scattering_c,
                                        float extinction_c, SegFrame* __restrict__ frames, float4* __restrict__ capA,
                                        float4* __restrict__ capB, HetFrame* __restrict__ het, HetFast* __restrict__ het_fast) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pb_segment s = segs[i];
  V3 start = v3(s.start[0], s.start[1], s.start[2]);
  V3 end = v3(s.end[0], s.end[1], s.end[2]);
  V3 ostart = v3(s.orig_start[0], s.orig_start[1], s.orig_start[2]);
  // HitCylinder, Intersections.h:313-328; HitCone (:151-175) builds the same frame about orig_start
  const bool cone = (mode == PB_MODE_GENERIC) && (s.flags & 1u);
  const V3 base = cone ? ostart : start;
  V3 difference = end - base;
  float len = length3(difference);
  V3 w_unit = normalize3(v3(0.0f, len, 0.0f));
  float Sinv = 1 / len;
  V3 c_unit = difference * Sinv;
  M3 Rinv = find_rotation_matrix(c_unit, w_unit);
  V3 Tinv = -(mul(Rinv, base) * Sinv);
  float rs = s.radius * Sinv;
  // float + float - double, rounded to float once (Intersections.h:344-346)
  float c = (float)((double)(Tinv.x * Tinv.x + Tinv.z * Tinv.z) - (double)rs * (double)rs);
  ConeFrame cf;
  cf.c2 = 1.0; cf.s2 = 0.0; cf.pad[0] = cf.pad[1] = cf.pad[2] = cf.pad[3] = 0.0f;
  float ymin = 0.0f;
  if (cone) {
    // :169-175, :196-200.  cosf/sinf of the host library are the correctly rounded values (to within a
    // rounding boundary met about once in 1e8 arguments); the double routines rounded to float give the same.
    const float tan_theta = rs / len;
    const float ct = (float)cos((double)tan_theta), st = (float)sin((double)tan_theta);
    cf.c2 = (double)ct * (double)ct;
    cf.s2 = (double)st * (double)st;
    c = (float)(cf.c2 * (double)(Tinv.x * Tinv.x + Tinv.z * Tinv.z) - (cf.s2 * (double)Tinv.y) * (double)Tinv.y);
    ymin = ((mul(Rinv, start) * Sinv) + Tinv).y;
  }
  M3 R = inverse3(Rinv);
  SegFrame f;
  for (int col = 0; col < 3; ++col)
    for (int row = 0; row < 3; ++row) { f.Rinv[col * 3 + row] = Rinv.m[col][row]; f.R[col * 3 + row] = R.m[col][row]; }
  f.ox = Tinv.x; f.oy = Tinv.y; f.oz = Tinv.z;
  f.c4 = 4 * c;
  // |seg.start - seg.orig_start| as a vec4 length, w = 1 - 1 = 0  (Intersections.h:440)
  V4 d4 = v4(start.x - ostart.x, start.y - ostart.y, start.z - ostart.z, 0.0f);
  f.tb0 = length4(d4);
  f.ox2 = 2 * Tinv.x; f.oz2 = 2 * Tinv.z;
  // beam_dir of the estimators is normalize(end - start) (BeamRadiance.h:64, :87) also where the frame is a cone's
  const V3 bdir = cone ? normalize3(end - start) : c_unit;
  f.bdir[0] = bdir.x; f.bdir[1] = bdir.y; f.bdir[2] = bdir.z;
  for (int k = 0; k < 3; ++k) { f.lo[k] = s.cull_lo[k]; f.hi[k] = s.cull_hi[k]; }
  gate_setup(f.lo, f.hi, f.cand, f.qflags);
  {
    // does the gate box contain the flat-capped cylinder?  Its extent along axis k about an end point is
    // r * sqrt(1 - dir_k^2); a few ulps of slack keep boxes that touch it (axis-parallel beams) in.
    bool contains = true;
    const float se[3][2] = {{start.x, end.x}, {start.y, end.y}, {start.z, end.z}};
    const float dk[3] = {c_unit.x, c_unit.y, c_unit.z};
    for (int k = 0; k < 3; ++k) {
      const float e = s.radius * pb_sqrt(fmaxf(0.0f, 1.0f - dk[k] * dk[k]));
      const float mn = fminf(se[k][0], se[k][1]) - e, mx = fmaxf(se[k][0], se[k][1]) + e;
      const float tol = 4e-6f * (1.0f + fmaxf(fabsf(mn), fabsf(mx)));
      contains = contains && (f.lo[k] <= mn + tol) && (f.hi[k] >= mx - tol);
    }
    if (contains && len > 0.0f && !cone) f.qflags |= 16u;
  }
  if (cone) f.qflags |= 32u;
  // BeamRadiance.h:446-455 (721) / :389-397 (73het): energy * phase_f [* rad]
  double phase_f = 1 / (4 * PB_PI_REF);
  for (int k = 0; k < 3; ++k) {
    double w = (double)s.energy[k] * phase_f;
    if (mode == PB_MODE_721 || (mode == PB_MODE_GENERIC && !cone && !one_dimensional)) {
      double rad = (double)scattering_c / (PB_PI_REF * ((double)s.radius * (double)s.radius));
      w = w * rad;
    }
    // the fixed-point sums take finite, non-negative terms: a zero radius or a negative or non-finite energy
    // (uploaded records) weighs nothing or saturates instead of corrupting its tile's sums
    if (!(w >= 0.0)) w = 0.0;
    f.W[k] = fmin(w, 1e30);
  }
  {
    const double wmax = fmax(f.W[0], fmax(f.W[1], f.W[2]));
    f.vcap10 = (float)fmin(PB_FIX_LIMIT * PB_FIX_PRESCALE / wmax, 3.0e38);   // wmax = 0: no cap needed, none overflows
  }
  if (f.W[0] == f.W[1] && f.W[1] == f.W[2]) f.qflags |= 64u;
  f.radius = s.radius;
  f.len = len;
  f.ymin = ymin;
  frames[i] = f;
  // Culling radius.  A cylinder lies within `radius` of its axis.  A cone's quadric (and its second nappe,
  // which HitCone does not exclude) can reach anywhere, but a contribution needs the ray to enter the
  // segment's gate box (BeamRadiance.h:476-479): an adaptive-width segment is culled by a capsule around
  // that box, in every mode, so that the bounding hierarchy does not depend on the estimator.
  float reach = s.radius;
  if (s.flags & 1u) reach = fmaxf(reach, box_reach(f.lo, f.hi, start, end));
  capA[i] = make_float4(start.x, start.y, start.z, reach);
  capB[i] = make_float4(end.x, end.y, end.z, 0.0f);
  HetFrame h;
  h.os[0] = ostart.x; h.os[1] = ostart.y; h.os[2] = ostart.z;
  h.s_ext = s.s_ext; h.e_ext = s.e_ext; h.c_ext = s.c_ext;
  h.seg_id = s.seg_id; h.radius = s.radius;
  if (cone) *reinterpret_cast<ConeFrame*>(&het[i]) = cf;
  else het[i] = h;
  if (het_fast && (mode == PB_MODE_73 || mode == PB_MODE_73HET)) {
    // orig_start in the cylinder's frame (start is its origin): R_inv (orig_start - start) S_inv
    const V3 q = mul(Rinv, ostart - start) * Sinv;
    HetFast hf;
    hf.r2 = q.x * q.x + q.z * q.z;
    hf.osy = q.y;
    hf.aos = pb_sqrt(hf.r2 + q.y * q.y);
    hf.bos = pb_sqrt(hf.r2 + (1.0f - q.y) * (1.0f - q.y));
    if (mode == PB_MODE_73HET) {
      // beam_extinction = c_ext + (s_ext + (e_ext - s_ext) proportion - c_ext) / (seg_id + 1), linear in proportion
      const float inv_seg = 1.0f / (float)(s.seg_id + 1);
      const float span = hf.bos - hf.aos;
      hf.A = s.c_ext + (s.s_ext - s.c_ext) * inv_seg;
      hf.B = span != 0.0f ? (s.e_ext - s.s_ext) * inv_seg / span : 0.0f;
    } else { hf.A = extinction_c; hf.B = 0.0f; }
    hf.krad = (scattering_c / extinction_c) / s.radius;
    hf.pad = 0.0f;
    het_fast[i] = hf;
  }
}

int pb_launch_prepare_scene(const float* d_tri_xyz, int n_tri, const float* d_matrix16, TriPrep* d_out,
                            float4* d_screen_box, float* d_screen_edge, hipStream_t s) {
  hipLaunchKernelGGL(prepare_scene_kernel, dim3((n_tri + 63) / 64), dim3(64), 0, s, d_tri_xyz, n_tri, d_matrix16, d_out,
                     d_screen_box, d_screen_edge);
  PB_HIP(hipGetLastError());
  return PB_OK;
}

int pb_launch_prepare_segments(const pb_segment* d_segs, int n, int mode, int one_dimensional, float scattering_c, float extinction_c,
                               SegFrame* d_frames, float4* d_capA, float4* d_capB, HetFrame* d_het, HetFast* d_het_fast, hipStream_t s) {
  if (n <= 0) return PB_OK;
  hipLaunchKernelGGL(prepare_segments_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_segs, n, mode, one_dimensional,
                     scattering_c, extinction_c, d_frames, d_capA, d_capB, d_het, d_het_fast);
  PB_HIP(hipGetLastError());
  return PB_OK;
}
