// The (ray, segment) pair evaluation the frame kernels are built from: HitBoundingBox gate, HitCylinder with
// the per-segment frame prepared, Integral_721 (exact-order and fast forms) and Integral_73Het.  Included by
// pb_render.hip (the kernels) and pb_probe.hip (one-lane-per-query access for known-answer tests); both are
// compiled with -ffp-contract=off.
#ifndef PB_PAIR_H
#define PB_PAIR_H
#include "pb_internal.h"
#include "pb_geom.h"
#include "pb_grid.h"
#include "pb_philox.h"

struct Ray {
  V3 l;        // normalize(vec3(dir))              Intersections.h:313, BeamRadiance.h:88
  V3 dn;       // xyz of normalize(vec4 dir) again  Intersections.h:91
  V3 d1;       // xyz of the vec4 dir itself (normalised once, skeleton.cpp:345-349): Integral_73Het's `dir`
  float maxd;  // |limit.position - start|          BeamRadiance.h:426
  bool active;
};

// HitBoundingBox (Intersections.h:81-143) for a ray from (0,0,0) followed by the
// distance test of BeamRadiance.h:429-430.  lo/hi are min_b/max_b, cand/qflags
// the origin-dependent part prepared per box.
PB_DI bool box_gate(const float lo[3], const float hi[3], const float cand[3], uint32_t qflags, V3 dn, float maxd,
                     float* entry_distance = nullptr) {
  if (qflags & 8u) { if (entry_distance) *entry_distance = 0.0f; return 0.0f <= maxd; }   // origin inside: hit = start, distance 0
  const float d[3] = {dn.x, dn.y, dn.z};
  float dist[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) dist[i] = (d[i] != 0 && !((qflags >> i) & 1u)) ? (cand[i] / d[i]) : -1.0f;
  int largest = 0;
  float dl = dist[0];
  if (dl < dist[1]) { largest = 1; dl = dist[1]; }
  if (dl < dist[2]) { largest = 2; dl = dist[2]; }
  if (dl < 0) return false;
  float pos[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (largest != i) {
      pos[i] = dl * d[i];                       // origin[i] + ... with origin 0
      if (pos[i] < lo[i] || pos[i] > hi[i]) return false;
    } else pos[i] = cand[i];
  }
  // |hit - start| as a vec4 length with w = 1 - 1 = 0
  const float hd = pb_sqrt((pos[0] * pos[0] + pos[1] * pos[1]) + (pos[2] * pos[2] + 0.0f));
  if (entry_distance) *entry_distance = hd;
  return hd <= maxd;
}

// The same decision from reciprocals: 1 = passes, 0 = fails, 2 = within 1e-5 (relative) of the distance
// threshold or of a box face, where only the exact-order form above may decide.
PB_DI int box_gate_quick(const float lo[3], const float hi[3], const float cand[3], uint32_t qflags, V3 dn, float maxd) {
  if (qflags & 8u) return (0.0f <= maxd) ? 1 : 0;
  const float d[3] = {dn.x, dn.y, dn.z};
  float dist[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) dist[i] = (d[i] != 0 && !((qflags >> i) & 1u)) ? cand[i] * __builtin_amdgcn_rcpf(d[i]) : -1.0f;
  const float dl = fmaxf(dist[0], fmaxf(dist[1], dist[2]));
  const float tol = 1e-5f * (1.0f + fabsf(dl));
  if (dl < -tol) return 0;
  if (dl < tol) return 2;
  // two candidate planes at nearly the same distance: which axis counts as the largest is the exact form's call
  const int near = (dist[0] > dl - tol) + (dist[1] > dl - tol) + (dist[2] > dl - tol);
  if (near > 1) return 2;
  bool inside = true, clear = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (dist[i] > dl - tol) continue;           // the entry face itself
    const float p = dl * d[i];
    inside = inside && (p >= lo[i] - tol) && (p <= hi[i] + tol);
    clear = clear && (p >= lo[i] + tol) && (p <= hi[i] - tol);
  }
  if (!inside) return 0;
  if (!clear) return 2;
  // |hit| = dl |dn| with |dn| = 1 to a few ulp
  if (dl < maxd * 0.99999f) return 1;
  if (dl > maxd * 1.00001f) return 0;
  return 2;
}

// Per-pixel set-up of DrawStandard, skeleton.cpp:342-350, then the root gate.
// tri_mask (wave-uniform): triangles that can be hit by any ray of the wave, see closest_intersection.
// tris / tri_box: where the prepared triangles are read from (RenderParams' own, or a copy of them in LDS).
template <class TriPtr, class BoxPtr>
PB_DI Ray make_ray(const RenderParams& p, int sx, int sy, unsigned long long tri_mask, TriPtr tris, BoxPtr tri_box) {
  Ray r;
  const float x_dir = sx - ((p.width * p.ssaa) / (float)2);
  const float y_dir = sy - ((p.height * p.ssaa) / (float)2);
  const V4 dir = normalize4(v4(x_dir, y_dir, p.focal, 1.0f));
  const V4 start = v4(0, 0, 0, 1);
  V4 pos = v4(0, 0, 0, 0);
  float dist; int idx;
  r.active = closest_intersection(tris, p.n_tri, start, dir, pos, dist, idx, tri_box, tri_mask);
  r.maxd = r.active ? length4(pos - start) : -1.0f;
  if (p.short_view && r.active) {
    // SHORT_VIEW (skeleton.cpp:442-449): the view ray ends at a free-flight distance drawn for it; the draw
    // is keyed by the sub-pixel, not by the order in which pixels are visited
    const double u = philox_canonical((uint32_t)sx, (uint32_t)sy, p.view_seed, PB_VIEW_KEY);
    const float t_a = (float)(-(log(1 - u) / (double)p.extinction_c));
    if (t_a > 0) {
      const float md = fminf(t_a, r.maxd);
      pos = start + (dir * md);
      r.maxd = length4(pos - start);          // BeamRadiance.h:474 on the moved limit
    }
  }
  r.l = normalize3(xyz(dir));
  r.dn = xyz(normalize4(dir));
  r.d1 = xyz(dir);
  return r;
}

PB_DI Ray make_ray(const RenderParams& p, int sx, int sy, unsigned long long tri_mask = ~0ull) {
  return make_ray(p, sx, sy, tri_mask, p.tris, p.tri_box);
}

// CylIntersection (Intersections.h:29-40) of a valid hit.
struct CylHit {
  V3 entry, exit;
  float tb_minus, tb_plus, tc_minus, tc_plus;
  bool first_inte;   // HitCylinder's return value (a first intersection exists), set on every path
};

// HitCylinder (Intersections.h:307-456) with the per-segment frame prepared, then the segment's gate
// box (BeamRadiance.h:426-430).  Returns 0 = no cylinder hit, 1 = discriminant >= 0 but nothing to
// integrate, 2 = valid intersection behind an open gate (h filled in).
PB_DI int hit_cylinder_exact(const SegFrame& f, const Ray& r, CylHit& h) {
  h.first_inte = false;
  const V3 l = r.l;
  const float dpx = f.Rinv[0] * l.x + f.Rinv[3] * l.y + f.Rinv[6] * l.z;
  const float dpz = f.Rinv[2] * l.x + f.Rinv[5] * l.y + f.Rinv[8] * l.z;
  const float a = dpx * dpx + dpz * dpz;
  const float b = f.ox2 * dpx + f.oz2 * dpz;
  const float b24ac = b * b - a * f.c4;
  if (b24ac < 0) return 0;
  const float sq = pb_sqrt(b24ac);
  float t0 = (-b + sq) / (2 * a);
  float t1 = (-b - sq) / (2 * a);
  if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
  const float dpy = f.Rinv[1] * l.x + f.Rinv[4] * l.y + f.Rinv[7] * l.z;
  const float ox = f.ox, oy = f.oy, oz = f.oz;
  const float y0 = oy + t0 * dpy;
  const float y1 = oy + t1 * dpy;
  float thf = 0.0f, ths = 0.0f;
  bool have_second = false;
  if (y0 < 0) {
    if (y1 < 0) return 1;
    thf = t0 + (t1 - t0) * y0 / (y0 - y1);
    if (thf <= 0) return 1;
    if (y1 <= 1) { ths = t1; have_second = true; }
    else { float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
  } else if (y0 >= 0 && y0 <= 1) {
    if (t0 <= 0) return 1;
    thf = t0;
    if (y1 >= 0) {
      if (y1 <= 1) { ths = t1; have_second = true; }
      else { float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
    } else { float th2 = t0 + (t1 - t0) * (y0) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
  } else if (y0 > 1) {
    if (y1 > 1) return 1;
    thf = t0 + (t1 - t0) * (y0 - 1) / (y0 - y1);
    if (thf <= 0) return 1;
    if (y1 >= 0) { ths = t1; have_second = true; }
    else { float th2 = t0 + (t1 - t0) * (y0) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
  } else return 1;
  h.first_inte = true;
  if (!((t0 > 0) && (t1 > 0) && (t1 > t0))) return 1;
  // the segment's gate box; the root gate was applied to the ray (maxd < 0 if it failed)
  if (!box_gate(f.lo, f.hi, f.cand, f.qflags, r.dn, r.maxd)) return 1;
  const float fx = ox + dpx * thf, fy = oy + dpy * thf, fz = oz + dpz * thf;
  // second_hit is unset in the reference when no branch assigned it; that never
  // coincides with valid == true, (0,0,0) stands in as in the CPU restatement
  const float sx = have_second ? ox + dpx * ths : 0.0f;
  const float sy = have_second ? oy + dpy * ths : 0.0f;
  const float sz = have_second ? oz + dpz * ths : 0.0f;
  const float Sinv = 1 / f.len;   // S_inv of Intersections.h:321: the division made when the record was built, again
  const float tb_a = f.tb0 + (fy / Sinv);
  const float tb_b = f.tb0 + (sy / Sinv);
  const float tb_minus = fminf(tb_a, tb_b);
  const float ux = fx - ox, uy = fy - oy, uz = fz - oz;
  const float ex = (f.R[0] * ux + f.R[3] * uy + f.R[6] * uz) / Sinv;
  const float ey = (f.R[1] * ux + f.R[4] * uy + f.R[7] * uz) / Sinv;
  const float ez = (f.R[2] * ux + f.R[5] * uy + f.R[8] * uz) / Sinv;
  const float wx = sx - ox, wy = sy - oy, wz = sz - oz;
  const float qx = (f.R[0] * wx + f.R[3] * wy + f.R[6] * wz) / Sinv;
  const float qy = (f.R[1] * wx + f.R[4] * wy + f.R[7] * wz) / Sinv;
  const float qz = (f.R[2] * wx + f.R[5] * wy + f.R[8] * wz) / Sinv;
  h.entry = v3(ex, ey, ez);
  h.exit = v3(qx, qy, qz);
  h.tb_minus = tb_minus;
  h.tb_plus = fmaxf(tb_a, tb_b);
  h.tc_minus = pb_sqrt(ex * ex + ey * ey + ez * ez);
  h.tc_plus = pb_sqrt(qx * qx + qy * qy + qz * qz);
  return 2;
}

// Integral_721 (BeamRadiance.h:86-110) on a valid hit: float products and float exp, widened for the
// quotient.  Same return convention as hit_cylinder_exact.
PB_DI double integral_721(const SegFrame& f, const CylHit& h, const Ray& r, float extinction) {
  const V3 l = r.l;
  const float cos_theta = f.bdir[0] * (-l.x) + f.bdir[1] * (-l.y) + f.bdir[2] * (-l.z);
  const float cm1 = fabsf(cos_theta) - 1;
  const float num = expf(-extinction * (h.tc_minus - h.tc_plus) * cm1) - 1;
  const float den = expf(extinction * (h.tc_minus + h.tb_minus)) * extinction * cm1;
  return (double)num / (double)den;
}

PB_DI int eval_pair_721(const SegFrame& f, const Ray& r, float extinction, double& integral_out) {
  CylHit h;
  const int res = hit_cylinder_exact(f, r, h);
  if (res != 2) return res;
  integral_out = integral_721(f, h, r, extinction);
  return 2;
}

// Integral_73Het (BeamRadiance.h:211-288) and, with no grid, Integral_73 (:112-209) for a homogeneous
// medium: closest approach of the beam axis and the view ray in
// double, the beam's extinction interpolated within the segment, the view ray's extinction by a
// walk through the grid from the camera to the point of closest approach.
struct D3v { double x, y, z; };
PB_DI D3v d3(V3 v) { D3v r; r.x = (double)v.x; r.y = (double)v.y; r.z = (double)v.z; return r; }
PB_DI D3v operator+(D3v a, D3v b) { D3v r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
PB_DI D3v operator-(D3v a, D3v b) { D3v r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
PB_DI D3v operator*(double s, D3v a) { D3v r; r.x = s * a.x; r.y = s * a.y; r.z = s * a.z; return r; }
PB_DI D3v operator*(D3v a, double s) { D3v r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
PB_DI double ddot(D3v a, D3v b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PB_DI double dlength(D3v a) { return __builtin_sqrt(ddot(a, a)); }
PB_DI D3v dnormalize(D3v a) { return a * (1.0 / __builtin_sqrt(ddot(a, a))); }

// The part of Integral_73Het that only the segment enters (BeamRadiance.h:218-219, :224, :252-254).
PB_DI BeamAxis beam_axis(V3 seg_start, V3 seg_end, const HetFrame& hf) {
  const D3v a = d3(seg_start), b = d3(seg_end);
  const D3v ab = dnormalize(b - a);
  const D3v os = d3(v3(hf.os[0], hf.os[1], hf.os[2]));
  BeamAxis ax;
  ax.ab[0] = ab.x; ax.ab[1] = ab.y; ax.ab[2] = ab.z;
  ax.u_dot_u = ddot(ab, ab);
  ax.a_os = dlength(a - os); ax.b_os = dlength(b - os);
  return ax;
}

PB_DI double integral_73het(const GridView& grid, V3 seg_start, const HetFrame& hf, const BeamAxis& ax, const CylHit& h,
                            const Ray& r, float extinction_c, float scattering_c) {
  const D3v a = d3(seg_start), c = d3(h.exit);
  const D3v d = d3(h.entry);
  D3v ab; ab.x = ax.ab[0]; ab.y = ax.ab[1]; ab.z = ax.ab[2];
  const D3v cd = dnormalize(d - c);
  const double u_dot_u = ax.u_dot_u, u_dot_v = ddot(ab, cd), v_dot_v = ddot(cd, cd);
  const double u_dot_w = ddot(ab, a - c), v_dot_w = ddot(cd, a - c);
  const double denominator = u_dot_u * v_dot_v - u_dot_v * u_dot_v;
  if (!((denominator != 0) && (u_dot_u != 0) && (v_dot_v != 0))) return 0.0;
  const double t_ab = (u_dot_v * v_dot_w - v_dot_v * u_dot_w) / denominator;
  const double t_cd = (u_dot_u * v_dot_w - u_dot_v * u_dot_w) / denominator;
  const D3v point_beam = a + (t_ab * ab);
  const D3v point_view = c + (t_cd * cd);
  const D3v os = d3(v3(hf.os[0], hf.os[1], hf.os[2]));
  const double view_distance = dlength(point_view);
  const double beam_distance = dlength(point_beam - os);
  const double a_os = ax.a_os, b_os = ax.b_os;
  const bool bound1 = (view_distance >= dlength(d)) && (view_distance <= dlength(c));
  const bool bound2 = (beam_distance >= a_os) && (beam_distance <= b_os);
  if (!(bound1 && bound2)) return 0.0;
  const double t_bc = beam_distance;
  const double t_cb = view_distance;
  double transmitted;
  float scattering = scattering_c;
  if (grid.z) {
    const float proportion = (float)((t_bc - a_os) / (b_os - a_os));
    const float additional = hf.s_ext + ((hf.e_ext - hf.s_ext) * proportion);
    const float beam_extinction = hf.c_ext + ((additional - hf.c_ext) / (hf.seg_id + 1));
    // vec4(0,0,0,1) + dir * (float) t_cb, dir the vec4 DrawHet normalised once (skeleton.cpp:383-387)
    const float reach = (float)t_cb;
    const V3 view_end = v3(0.0f + (r.d1.x * reach), 0.0f + (r.d1.y * reach), 0.0f + (r.d1.z * reach));
    const float view_extinction = extinction3d_mean(grid, v3(0, 0, 0), view_end);
    transmitted = exp(-(double)beam_extinction * t_bc) * exp(-(double)view_extinction * t_cb);
    scattering = (beam_extinction / extinction_c) * scattering_c;
  } else {
    // Integral_73 in a homogeneous medium (BeamRadiance.h:189-192): sigma_t along both paths
    transmitted = exp(-(double)extinction_c * t_bc) * exp(-(double)extinction_c * t_cb);
  }
  if (transmitted < 1e-6) return 0.0;
  const double cos_theta = ddot(ab, cd);
  const double sin_theta = __builtin_sqrt(1 - cos_theta * cos_theta);
  const double rad = (double)(scattering / hf.radius);
  return transmitted * rad / sin_theta;
}

PB_DI double integral_73het(const GridView& grid, V3 seg_start, V3 seg_end, const HetFrame& hf, const CylHit& h,
                            const Ray& r, float extinction_c, float scattering_c) {
  return integral_73het(grid, seg_start, hf, beam_axis(seg_start, seg_end, hf), h, r, extinction_c, scattering_c);
}

// One (ray, segment) pair of BVHBeamRadianceHet (BeamRadiance.h:378-398).
PB_DI int eval_pair_73het(const SegFrame& f, const HetFrame& hf, V3 seg_start, V3 seg_end, const GridView& grid,
                          const Ray& r, float extinction_c, float scattering_c, double& integral_out) {
  CylHit h;
  const int res = hit_cylinder_exact(f, r, h);
  if (res != 2) return res;
  integral_out = integral_73het(grid, seg_start, seg_end, hf, h, r, extinction_c, scattering_c);
  return 2;
}

// HitCone (Intersections.h:145-305) with the per-segment frame prepared (about orig_start), then the segment's
// gate box.  Same return convention as hit_cylinder_exact.  pow( float, 2 ) promotes: the coefficients are
// double expressions of the float terms, rounded to float once.
PB_DI int hit_cone_exact(const SegFrame& f, const ConeFrame& cf, const Ray& r, CylHit& h) {
  h.first_inte = false;
  const V3 l = r.l;
  const float dpx = f.Rinv[0] * l.x + f.Rinv[3] * l.y + f.Rinv[6] * l.z;
  const float dpy = f.Rinv[1] * l.x + f.Rinv[4] * l.y + f.Rinv[7] * l.z;
  const float dpz = f.Rinv[2] * l.x + f.Rinv[5] * l.y + f.Rinv[8] * l.z;
  const float ox = f.ox, oy = f.oy, oz = f.oz, ymin = f.ymin;
  const float a = (float)(cf.c2 * (double)(dpx * dpx + dpz * dpz) - (cf.s2 * (double)dpy) * (double)dpy);
  const float b = (float)(cf.c2 * (double)(f.ox2 * dpx + f.oz2 * dpz) - ((cf.s2 * 2.0) * (double)oy) * (double)dpy);
  const float b24ac = b * b - a * f.c4;
  if (b24ac < 0) return 0;
  const float sq = pb_sqrt(b24ac);
  float t0 = (-b + sq) / (2 * a);
  float t1 = (-b - sq) / (2 * a);
  if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
  const float y0 = oy + t0 * dpy;
  const float y1 = oy + t1 * dpy;
  float thf = 0.0f, ths = 0.0f;
  bool have_second = false;
  if (y0 < ymin) {
    if (y1 < ymin) return 1;
    thf = t0 + (t1 - t0) * (y0 - ymin) / (y0 - y1);
    if (thf <= 0) return 1;
    if (y1 <= 1) { ths = t1; have_second = true; }
    else { float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
  } else if (y0 >= ymin && y0 <= 1) {
    if (t0 <= 0) return 1;
    thf = t0;
    if (y1 >= ymin) {
      if (y1 <= 1) { ths = t1; have_second = true; }
      else { float th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
    } else { float th2 = t0 + (t1 - t0) * (y0 - ymin) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
  } else if (y0 > 1) {
    if (y1 > 1) return 1;
    thf = t0 + (t1 - t0) * (y0 - 1) / (y0 - y1);
    if (thf <= 0) return 1;
    if (y1 >= 0) { ths = t1; have_second = true; }                  // :275 compares with 0, not with min
    else { float th2 = t0 + (t1 - t0) * (y0 - ymin) / (y0 - y1); if (th2 > 0) { ths = th2; have_second = true; } }
  } else return 1;
  h.first_inte = true;
  if (!((t0 > 0) && (t1 > 0) && (t1 > t0))) return 1;
  if (!box_gate(f.lo, f.hi, f.cand, f.qflags, r.dn, r.maxd)) return 1;
  const float fx = ox + dpx * thf, fy = oy + dpy * thf, fz = oz + dpz * thf;
  const float sx = have_second ? ox + dpx * ths : 0.0f;
  const float sy = have_second ? oy + dpy * ths : 0.0f;
  const float sz = have_second ? oz + dpz * ths : 0.0f;
  const float Sinv = 1 / f.len;   // S_inv of Intersections.h:158, as above
  const float ux = fx - ox, uy = fy - oy, uz = fz - oz;
  const float ex = (f.R[0] * ux + f.R[3] * uy + f.R[6] * uz) / Sinv;
  const float ey = (f.R[1] * ux + f.R[4] * uy + f.R[7] * uz) / Sinv;
  const float ez = (f.R[2] * ux + f.R[5] * uy + f.R[8] * uz) / Sinv;
  const float wx = sx - ox, wy = sy - oy, wz = sz - oz;
  const float qx = (f.R[0] * wx + f.R[3] * wy + f.R[6] * wz) / Sinv;
  const float qy = (f.R[1] * wx + f.R[4] * wy + f.R[7] * wz) / Sinv;
  const float qz = (f.R[2] * wx + f.R[5] * wy + f.R[8] * wz) / Sinv;
  h.entry = v3(ex, ey, ez);
  h.exit = v3(qx, qy, qz);
  h.tb_minus = f.tb0 + (fy / Sinv);                               // :297-298: no ordering of the two
  h.tb_plus = f.tb0 + (sy / Sinv);
  h.tc_minus = pb_sqrt(ex * ex + ey * ey + ez * ez);
  h.tc_plus = pb_sqrt(qx * qx + qy * qy + qz * qz);
  return 2;
}

// Integral_721_ada (BeamRadiance.h:63-84): the Riemann sum in steps of 0.001 along the view ray, in double.
// f.len of a cone record is |end - orig_start|, the reference's length_const.  The step count is bounded
// (the reference's loop is not): a hit whose exit distance overflowed must not keep a wave running.
PB_DI double integral_721_ada(const SegFrame& f, const CylHit& h, const Ray& r, float extinction, float scattering_c) {
  const V3 l = r.l;
  const float cos_f = f.bdir[0] * (-l.x) + f.bdir[1] * (-l.y) + f.bdir[2] * (-l.z);
  const double ac = __builtin_fabs((double)cos_f);
  const double width = (double)f.radius / (double)f.len;
  const double ext = (double)extinction, sc = (double)scattering_c;
  const double tcm = (double)h.tc_minus, tcp = (double)h.tc_plus, tbm = (double)h.tb_minus;
  const double dt_c = 0.001;
  double integrand = 0.0;
  int guard = 0;
  for (double tc = tcm; tc < tcp && guard < 1000000; tc = tc + dt_c, ++guard) {
    const double tb = tbm - (ac * (tc - tcm));
    const double current_r = width * tb;
    const double constant = exp(-ext * tb);
    const double transmitted = exp(-ext * tc) * constant * (sc / (current_r * current_r)) * dt_c;
    integrand += transmitted;
  }
  return integrand;
}

// One (ray, segment) pair of BeamRadiance (BeamRadiance.h:486-543): a cone record goes through HitCone and
// Integral_721_ada, any other through HitCylinder and Integral_73 (ONE_DIMENSIONAL; Integral_73Het's
// arithmetic when a grid is given, :159-188) or Integral_721.
PB_DI int eval_pair_generic(const SegFrame& f, const HetFrame& hf, V3 seg_start, V3 seg_end, const GridView& grid,
                            bool one_dimensional, const Ray& r, float extinction_c, float scattering_c,
                            double& integral_out) {
  if (f.qflags & 32u) {
    ConeFrame cf;
    __builtin_memcpy(&cf, &hf, sizeof(cf));
    CylHit h;
    const int res = hit_cone_exact(f, cf, r, h);
    if (res != 2) return res;
    integral_out = integral_721_ada(f, h, r, extinction_c, scattering_c);
    return 2;
  }
  if (one_dimensional) return eval_pair_73het(f, hf, seg_start, seg_end, grid, r, extinction_c, scattering_c, integral_out);
  return eval_pair_721(f, r, extinction_c, integral_out);
}

// The same pair evaluation with the arithmetic the estimate is continuous in relaxed to what the
// hardware does fastest; the radiance differs from the exact-order form above by the reference's own
// float rounding noise (about 1e-5 relative per pixel), far inside the stated tolerance.
//  * fused multiply-adds, v_rcp/v_sqrt instead of IEEE divide/sqrt;
//  * validity first: a contributing pair needs 0 < t0 < t1 (Intersections.h:437), and then every cap
//    crossing lies in [t0, t1], so the th <= 0 tests of the y-cases (:366-432) cannot fire;
//  * the y-cases without branches: the hit points' y are y0, y1 clamped to [0, 1], and the ray parameter
//    where y reaches a cap is t0 + (t1 - t0) (y0 - cap) / (y0 - y1), the formula of all four cap branches;
//  * entry/exit distances from the ray parameter: the rotation back to world space and its length
//    (Intersections.h:443-450) return |l * th / S_inv| = th * |end - start| for a camera at the origin;
//  * cos(theta) = -dot(beam_dir, l) is the y component of R_inv * l;
//  * Integral_721 = (exp(x) - 1) / (exp(E) sigma_t (c - 1)) with x = sigma_t (tc+ - tc-) (c - 1) is evaluated as
//    (tc+ - tc-) * ((exp(x) - 1) / x) * exp(-E): no division, no 0/0 at c = 1, and (exp(x) - 1) / x by its
//    series where |x| is small (the reference's float exp loses 4 digits there);
//  * the gate: a segment whose gate box contains its cylinder (qflags bit 4) passes whenever the ray
//    enters the cylinder in front of the surface, since it enters the box no later; otherwise, and for
//    every other segment, the exact HitBoundingBox gate runs.
PB_DI float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// a * b with 0 * anything = 0 (also 0 * inf and 0 * NaN)
// The s_nop is the wait state gfx950 needs between a transcendental instruction (b is often a fresh v_rcp_f32) and
// an ordinary vector instruction that reads its result: the compiler inserts it for instructions it emits itself
// and does not look inside inline assembly.
PB_DI float mul_zero_wins(float a, float b) {
  float r;
  asm("s_nop 0\n\tv_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// The part of the fast evaluation that reads only what HitCylinder derives from the segment (the first sixteen
// dwords of its record) and the ray: 0 = no cylinder hit, 1 = nothing to integrate, 2 = a valid crossing whose
// entry distance is tc_minus and whose integral (times 2^log2_scale, folded into the exponential at no cost) is
// `value`, still to be passed by the segment's gate.
// The crossing's quadratic (the part every (ray, survivor) pair runs): returns the discriminant, negative = no hit.
PB_DI float pair_721_fast_quadratic(const float Rinv[9], float ox2, float oz2, float c4, V3 l, float& a_out, float& b_out) {
  const float dpx = __builtin_fmaf(Rinv[6], l.z, __builtin_fmaf(Rinv[3], l.y, Rinv[0] * l.x));
  const float dpz = __builtin_fmaf(Rinv[8], l.z, __builtin_fmaf(Rinv[5], l.y, Rinv[2] * l.x));
  const float a = __builtin_fmaf(dpx, dpx, dpz * dpz);
  const float b = __builtin_fmaf(ox2, dpx, oz2 * dpz);
  a_out = a; b_out = b;
  return __builtin_fmaf(b, b, -(a * c4));
}

// What follows the quadratic for a pair whose discriminant is not negative: ry = the middle row of R_inv (Rinv[1],
// Rinv[4], Rinv[7]).  1 = nothing to integrate, 2 = a valid crossing (see pair_721_fast_geometry).
PB_DI int pair_721_fast_crossing(float a, float b, float b24ac, V3 ry, float oy, float len, float tb0, V3 l,
                                 float extinction, float log2_scale, float& tc_minus_out, float& value);

PB_DI int pair_721_fast_geometry(const float Rinv[9], float ox2, float oz2, float c4, float oy, float len, float tb0, const Ray& r,
                                 float extinction, float log2_scale, float& tc_minus_out, float& value) {
  float a, b;
  const float b24ac = pair_721_fast_quadratic(Rinv, ox2, oz2, c4, r.l, a, b);
  if (b24ac < 0) return 0;
  return pair_721_fast_crossing(a, b, b24ac, v3(Rinv[1], Rinv[4], Rinv[7]), oy, len, tb0, r.l, extinction, log2_scale, tc_minus_out, value);
}

// The same with what is done with a valid crossing handed in as `then(tc_minus, value)`: called at the innermost point,
// so that nothing the crossing computes has to be given a value on the paths that leave early (in the pair loop those
// were eight register moves a step).
template <class Then>
PB_DI void pair_721_fast_crossing_then(float a, float b, float b24ac, V3 ry, float oy, float len, float tb0, V3 l,
                                       float extinction, float log2_scale, Then&& then);

PB_DI int pair_721_fast_crossing(float a, float b, float b24ac, V3 ry, float oy, float len, float tb0, V3 l,
                                 float extinction, float log2_scale, float& tc_minus_out, float& value) {
  int res = 1;
  pair_721_fast_crossing_then(a, b, b24ac, ry, oy, len, tb0, l, extinction, log2_scale,
                              [&](float tc_minus, float v) { tc_minus_out = tc_minus; value = v; res = 2; });
  return res;
}

template <class Then>
PB_DI void pair_721_fast_crossing_then(float a, float b, float b24ac, V3 ry, float oy, float len, float tb0, V3 l,
                                       float extinction, float log2_scale, Then&& then) {
  const float sq = __builtin_amdgcn_sqrtf(b24ac);
  const float inv2a = fast_rcp(a + a);
  const float t0 = (-b - sq) * inv2a;     // a > 0: already ordered, t0 <= t1
  const float t1 = (-b + sq) * inv2a;
  if (!((t0 > 0) && (t1 > t0))) return;
  const float dpy = __builtin_fmaf(ry.z, l.z, __builtin_fmaf(ry.y, l.y, ry.x * l.x));
  const float y0 = __builtin_fmaf(t0, dpy, oy);
  const float y1 = __builtin_fmaf(t1, dpy, oy);
  const float y_low = fminf(y0, y1);
  if (fmaxf(y0, y1) < 0.0f || y_low > 1.0f) return;                 // passes below or above both caps
  const float fy = __builtin_amdgcn_fmed3f(y0, 0.0f, 1.0f);         // y of the entry and exit points
  const float sy = __builtin_amdgcn_fmed3f(y1, 0.0f, 1.0f);
  const float q = -fast_rcp(dpy);                                   // (t1 - t0) / (y0 - y1); selected away where dpy == 0
  // t0 + (y0 - y) q, with a multiply for which 0 * inf = 0: where no cap is crossed the first factor is 0
  const float thf = t0 + mul_zero_wins(y0 - fy, q);
  const float ths = t1 + mul_zero_wins(y1 - sy, q);
  const float tc_minus = thf * len;
  const float tc_plus = ths * len;
  const float chord = tc_plus - tc_minus;
  // tc- + tb- = len (th_first + min(fy, sy)) + tb0, and min(fy, sy) is the clamped min(y0, y1)
  const float reach = __builtin_fmaf(thf + __builtin_amdgcn_fmed3f(y_low, 0.0f, 1.0f), len, tb0);
  const float cm1 = fabsf(dpy) - 1.0f;
  const float x = extinction * chord * cm1;
  float phi;                                                        // (exp(x) - 1) / x
  if (__builtin_expect(fabsf(x) < 0.02f, 1))
    phi = __builtin_fmaf(x, __builtin_fmaf(x, __builtin_fmaf(x, 1.0f / 24.0f, 1.0f / 6.0f), 0.5f), 1.0f);
  else phi = (__expf(x) - 1.0f) * fast_rcp(x);
  const float fall = __builtin_amdgcn_exp2f(__builtin_fmaf(reach, extinction * -1.44269504f, log2_scale));   // exp(-sigma_t (tc- + tb-))
  then(tc_minus, chord * phi * fall);
}

// The gate of a valid crossing: a segment whose gate box contains its cylinder (qflags bit 4) passes whenever the
// ray enters the cylinder in front of the surface; otherwise the box decides - its entry distance first with
// v_rcp, the exact-order gate only where that lands within 1e-5 of a threshold.
PB_DI bool pair_721_fast_sure(uint32_t qflags, float tc_minus, float maxd) { return (qflags & 16u) && (tc_minus < maxd * 0.99999f); }
PB_DI bool pair_721_fast_gate(const float lo[3], const float hi[3], const float cand[3], uint32_t qflags, const Ray& r) {
  const int quick = box_gate_quick(lo, hi, cand, qflags, r.dn, r.maxd);
  if (quick == 0) return false;
  if (quick == 2 && !box_gate(lo, hi, cand, qflags, r.dn, r.maxd)) return false;
  return true;
}

// The cylinder crossing for the relaxed-arithmetic Integral_73 / Integral_73Het, from the first sixteen dwords of the
// segment's record.  In the cylinder's frame the view ray's closest approach to the axis is at the mid-point of the
// two roots, t_mid = -b / 2a, where the axis point is y_star (in units of the segment's length from its start).
// That is the pair of points Integral_73Het finds by solving for the closest approach of the two lines in double
// (BeamRadiance.h:224-243); its bound on the view distance (:247, between the entry and the exit point) reads
// th_first <= t_mid <= th_second here.  sin^2(theta) between beam and view ray is a = dpx^2 + dpz^2 (R_inv is a
// rotation and l a unit vector), free of the cancellation in 1 - cos^2.
// This estimator is a box across the beam: a pair contributes in full or not at all, so a bound decided the other
// way than the reference decides it costs a whole term.  Where the decision is not safe in float arithmetic the pair
// is handed to the exact-order code instead (return value 3):
//  * the discriminant is computed with the operations of hit_cylinder_exact, so a cylinder is hit here exactly when
//    it is hit there;
//  * a beam seen nearly end-on (a < HET_FAST_MIN_SIN2): the closest approach is ill-conditioned and 1 / sin(theta) large;
//  * t_mid within HET_FAST_EDGE (relative) of th_first or th_second, or within what the rounding of a short and
//    distant segment's coordinates can move them (y_err_out: the same margin for the axis point's y).
// Returns 0 = no cylinder hit, 1 = nothing to integrate, 2 = a crossing whose closest approach lies inside it, 3 = undecided.
constexpr float HET_FAST_MIN_SIN2 = 0.01f;
constexpr float HET_FAST_EDGE = 5e-5f;
PB_DI int pair_het_fast_geometry(const float Rinv[9], float ox2, float oz2, float c4, float oy, float len, const Ray& r,
                                 float& tc_minus_out, float& t_mid_out, float& y_star_out, float& y_err_out, float& sin2_out) {
  const V3 l = r.l;
  const float dpx = Rinv[0] * l.x + Rinv[3] * l.y + Rinv[6] * l.z;
  const float dpz = Rinv[2] * l.x + Rinv[5] * l.y + Rinv[8] * l.z;
  const float a = dpx * dpx + dpz * dpz;
  const float b = ox2 * dpx + oz2 * dpz;
  const float b24ac = b * b - a * c4;
  if (b24ac < 0) return 0;
  if (!(a >= HET_FAST_MIN_SIN2)) return 3;
  const float sq = __builtin_amdgcn_sqrtf(b24ac);
  const float inv2a = fast_rcp(a + a);
  // the roots about their mid-point (which the estimator needs anyway)
  const float t_mid = -b * inv2a;
  const float half = sq * inv2a;
  const float t0 = t_mid - half;
  const float t1 = t_mid + half;
  if (!((t0 > 0) && (t1 > t0))) return 1;
  const float dpy = __builtin_fmaf(Rinv[7], l.z, __builtin_fmaf(Rinv[4], l.y, Rinv[1] * l.x));
  const float y0 = __builtin_fmaf(t0, dpy, oy);
  const float y1 = __builtin_fmaf(t1, dpy, oy);
  // y is the difference of two numbers that are large where the segment is short against its distance from the
  // camera (the frame's unit is the segment's length): its rounding error, generously
  const float y_err = 2e-6f * (fabsf(oy) + fabsf(t1 * dpy));
  if (fmaxf(y0, y1) < -y_err || fminf(y0, y1) > 1.0f + y_err) return 1;
  const float fy = __builtin_amdgcn_fmed3f(y0, 0.0f, 1.0f);
  const float sy = __builtin_amdgcn_fmed3f(y1, 0.0f, 1.0f);
  const float q = -fast_rcp(dpy);
  const float thf = t0 + mul_zero_wins(y0 - fy, q);
  const float ths = t1 + mul_zero_wins(y1 - sy, q);
  tc_minus_out = thf * len;
  // The reference takes the view ray's direction from the entry and exit points it has rounded to float: over a chord
  // of relative length (ths - thf) / t_mid that direction is off by about 1e-7 t_mid / (ths - thf), which moves the
  // closest approach by that angle over sin^2(theta) - for a sliver of a segment far more than any fixed margin.
  const float noise = 2.5e-7f * t_mid * fast_rcp(fmaxf(ths - thf, 1e-30f) * a);
  const float edge = __builtin_fmaf(fmaxf(HET_FAST_EDGE, noise), t_mid, y_err * fabsf(q));
  y_err_out = y_err + edge;                              // y moves by at most |dpy| <= 1 per unit of t
  const float past_first = t_mid - thf, short_of_second = ths - t_mid;      // both positive inside the crossing
  if (past_first < -edge || short_of_second < -edge) return 1;
  if (past_first < edge || short_of_second < edge) return 3;
  t_mid_out = t_mid;
  y_star_out = __builtin_fmaf(t_mid, dpy, oy);
  sin2_out = a;
  return 2;
}

// pair_het_fast_geometry without branches: every lane runs every stage (as soon as one lane of a wave passes a stage the
// wave issues it anyway; the nested exits only add execution-mask bookkeeping) and the outcome is put together from
// predicates at the end.  The arithmetic of a lane that passes a stage is the same, operation for operation, so the
// outcome is the same.  ok: a crossing whose closest approach lies inside it (return value 2 there); defer: undecided (3).
PB_DI void pair_het_fast_geometry_flat(const float Rinv[9], float ox2, float oz2, float c4, float oy, float len, const Ray& r,
                                       bool& ok_out, bool& defer_out, float& tc_minus_out, float& t_mid_out, float& y_star_out,
                                       float& y_err_out, float& sin2_out) {
  const V3 l = r.l;
  const float dpx = Rinv[0] * l.x + Rinv[3] * l.y + Rinv[6] * l.z;
  const float dpz = Rinv[2] * l.x + Rinv[5] * l.y + Rinv[8] * l.z;
  const float a = dpx * dpx + dpz * dpz;
  const float b = ox2 * dpx + oz2 * dpz;
  const float b24ac = b * b - a * c4;
  const bool hit = !(b24ac < 0);
  const bool end_on = !(a >= HET_FAST_MIN_SIN2);
  const float sq = __builtin_amdgcn_sqrtf(b24ac);
  const float inv2a = fast_rcp(a + a);
  const float t_mid = -b * inv2a;
  const float half = sq * inv2a;
  const float t0 = t_mid - half;
  const float t1 = t_mid + half;
  const bool ordered = ((int)(t0 > 0) & (int)(t1 > t0)) != 0;
  const float dpy = __builtin_fmaf(Rinv[7], l.z, __builtin_fmaf(Rinv[4], l.y, Rinv[1] * l.x));
  const float y0 = __builtin_fmaf(t0, dpy, oy);
  const float y1 = __builtin_fmaf(t1, dpy, oy);
  const float y_err = 2e-6f * (fabsf(oy) + fabsf(t1 * dpy));
  const bool in_caps = ((int)(fmaxf(y0, y1) < -y_err) | (int)(fminf(y0, y1) > 1.0f + y_err)) == 0;
  const float fy = __builtin_amdgcn_fmed3f(y0, 0.0f, 1.0f);
  const float sy = __builtin_amdgcn_fmed3f(y1, 0.0f, 1.0f);
  const float q = -fast_rcp(dpy);
  const float thf = t0 + mul_zero_wins(y0 - fy, q);
  const float ths = t1 + mul_zero_wins(y1 - sy, q);
  tc_minus_out = thf * len;
  const float noise = 2.5e-7f * t_mid * fast_rcp(fmaxf(ths - thf, 1e-30f) * a);
  const float edge = __builtin_fmaf(fmaxf(HET_FAST_EDGE, noise), t_mid, y_err * fabsf(q));
  y_err_out = y_err + edge;
  const float past_first = t_mid - thf, short_of_second = ths - t_mid;
  const bool outside = ((int)(past_first < -edge) | (int)(short_of_second < -edge)) != 0;
  const bool near = ((int)(past_first < edge) | (int)(short_of_second < edge)) != 0;
  const bool base = ((int)hit & (int)!end_on & (int)ordered & (int)in_caps & (int)!outside) != 0;
  ok_out = ((int)base & (int)!near) != 0;
  defer_out = ((int)hit & ((int)end_on | ((int)base & (int)near))) != 0;
  t_mid_out = t_mid;
  y_star_out = __builtin_fmaf(t_mid, dpy, oy);
  sin2_out = a;
}

// log2_scale: the integral is returned times 2^log2_scale (folded into the exponential at no cost).
template <bool CHECK>
PB_DI int eval_pair_721_fast(const SegFrame& f, const Ray& r, float extinction, double& integral_out, unsigned long long& shortcut_wrong,
                             float log2_scale = 0.0f) {
  float tc_minus, value;
  const int res = pair_721_fast_geometry(f.Rinv, f.ox2, f.oz2, f.c4, f.oy, f.len, f.tb0, r, extinction, log2_scale, tc_minus, value);
  if (res != 2) return res;
  if (!pair_721_fast_sure(f.qflags, tc_minus, r.maxd)) {
    if (!pair_721_fast_gate(f.lo, f.hi, f.cand, f.qflags, r)) return 1;
  } else if (CHECK) {
    if (!box_gate(f.lo, f.hi, f.cand, f.qflags, r.dn, r.maxd)) shortcut_wrong++;
  }
  integral_out = (double)value;
  return 2;
}

#endif
