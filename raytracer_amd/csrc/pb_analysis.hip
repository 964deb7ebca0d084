// RunAnalysis (Analysis.h:81-122) on the device: the normalised standard deviation (NSD.csv) and relative
// RMSE (rRMSE.csv) of the reference's photon point/beam estimators for its fixed study geometry, one lane per
// kernel width.  A row's cells share three intersections — the view ray with the beam's blur cylinder, the
// view ray with the blur sphere about the beam's point of closest approach, the beam with the sphere about
// the view ray's — which are found once per row; the reference recomputes them per cell.  All in double
// except where the reference itself narrows (findRotationMatrix in float, sqrtf on doubles).
#include <cmath>
#include <vector>

#include "pb_internal.h"

namespace {

struct P3 { double x, y, z; };
PB_HDI P3 p3(double x, double y, double z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
PB_HDI P3 operator+(P3 a, P3 b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }
PB_HDI P3 operator-(P3 a, P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
PB_HDI P3 operator*(P3 a, double s) { return p3(a.x * s, a.y * s, a.z * s); }
PB_HDI P3 operator*(double s, P3 a) { return p3(s * a.x, s * a.y, s * a.z); }
PB_HDI P3 operator/(P3 a, double s) { return p3(a.x / s, a.y / s, a.z / s); }
PB_HDI double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PB_DI double len(P3 a) { return __builtin_sqrt(dot(a, a)); }
PB_DI double sq(double x) { return x * x; }
PB_DI double sqrt_f(double x) { return (double)__builtin_sqrtf((float)x); }

struct M3d { double m[3][3]; };    // m[col][row]
PB_DI P3 mul(const M3d& m, P3 v) {
  return p3(m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z, m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
            m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z);
}

struct Geometry {            // SetUpAnalysis, Analysis.h:455-475
  double ext, mfp, t_a, t_c;
  P3 a, c, omega_a, omega_c;
  M3d R_inv, R;              // findRotationMatrix(beam axis -> +y) widened, and its inverse (:549, :649)
};

struct Span { P3 entry, exit; bool hit; };

// SphereIntersection, Analysis.h:487-528
PB_DI Span sphere(P3 p, P3 dir, double w, P3 centre) {
  Span s;
  s.entry = p3(0, 0, 0); s.exit = p3(0, 0, 0); s.hit = false;
  const P3 vpc = centre - p;
  const double along = dot(vpc, dir);
  const double lv = len(vpc);
  const double cos_theta = along / (lv * len(dir));
  const double distance = cos_theta * lv;
  const P3 pc = p + (distance * dir);
  if (along < 0) {
    if (lv > w) return s;
    if (lv == w) { s.entry = p; s.exit = p + ((2 * w) * dir); }
    else {
      const double dist = __builtin_sqrt(sq(w) - sq(len(pc - centre)));
      const double di1 = dist - len(pc - p);
      s.exit = p + (dir * di1);
      s.entry = p + (dir * di1) - (dir * (w * 2));
    }
  } else {
    if (len(centre - pc) > w) return s;
    const double dist = __builtin_sqrt(sq(w) - sq(len(pc - centre)));
    const double di1 = (lv > w) ? len(pc - p) - dist : len(pc - p) + dist;
    s.entry = p + (dir * di1);
    const double di2 = di1 + (2 * len(pc - s.entry));
    s.exit = p + (dir * di2);
  }
  s.hit = true;
  return s;
}

// CylinderIntersection, Analysis.h:530-653: the view ray against the cylinder of radius w about the beam
// (10 mean free paths long).
PB_DI Span cylinder(const Geometry& g, double w) {
  Span s;
  s.entry = p3(0, 0, 0); s.exit = p3(0, 0, 0); s.hit = false;
  const double S_inv = 1 / len((g.a + ((g.omega_a * 10.0) * g.mfp)) - g.a);
  const P3 T_inv = p3(0, 0, 0) - (mul(g.R_inv, g.a) * S_inv);
  const P3 o = mul(g.R_inv, g.c * S_inv) + T_inv;
  const P3 d = mul(g.R_inv, g.omega_c);
  const double aq = d.x * d.x + d.z * d.z;
  const double bq = 2 * o.x * d.x + 2 * o.z * d.z;
  const double cq = o.x * o.x + o.z * o.z - sq(w * S_inv);
  const double b24ac = bq * bq - 4 * aq * cq;
  if (b24ac < 0) return s;
  const double root = sqrt_f(b24ac);
  double t0 = (-bq + root) / (2 * aq), t1 = (-bq - root) / (2 * aq);
  if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
  const double y0 = o.y + t0 * d.y, y1 = o.y + t1 * d.y;
  double tf, ts = 0.0;
  bool second = false;
  if (y0 < 0) {
    if (y1 < 0) return s;
    tf = t0 + (t1 - t0) * y0 / (y0 - y1);
    if (tf <= 0) return s;
    if (y1 <= 1) { ts = t1; second = true; }
    else { const double th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) { ts = th2; second = true; } }
  } else if (y0 <= 1) {
    if (t0 <= 0) return s;
    tf = t0;
    if (y1 >= 0) {
      if (y1 <= 1) { ts = t1; second = true; }
      else { const double th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) { ts = th2; second = true; } }
    } else { const double th2 = t0 + (t1 - t0) * (y0) / (y0 - y1); if (th2 > 0) { ts = th2; second = true; } }
  } else if (y0 > 1) {
    if (y1 > 1) return s;
    tf = t0 + (t1 - t0) * (y0 - 1) / (y0 - y1);
    if (tf <= 0) return s;
    if (y1 >= 0) { ts = t1; second = true; }
    else { const double th2 = t0 + (t1 - t0) * (y0) / (y0 - y1); if (th2 > 0) { ts = th2; second = true; } }
  } else return s;               // NaN
  const P3 first = o + (d * tf);
  const P3 last = second ? o + (d * ts) : p3(0, 0, 0);     // unset in the reference; zero as in the CPU restatement
  s.entry = mul(g.R, first - T_inv) / S_inv;
  s.exit = mul(g.R, last - T_inv) / S_inv;
  s.hit = true;
  return s;
}

struct Row {
  const Geometry& g;
  double w;
  PB_DI double Tr(double t) const { return exp(-g.ext * t); }                                    // :477-480
  PB_DI double TrP(double t0, double t1) const { return (Tr(t0) - Tr(t1)) / g.ext; }            // :482-485
  PB_DI double wpow(int dim) const { return pow(w, (double)(-dim)); }
  // second moment of an estimate blurred over [t0, t1] of one path, the other path contributing `other`
  // (:345-347, :442-444)
  PB_DI double blurred(double other, double t0, double t1) const {
    return other * (2 / sq(g.ext)) * (Tr(t0) + (Tr(t1) * (((t0 - t1) * g.ext) - 1)));
  }
};

__global__ __launch_bounds__(64) void analysis_kernel(Geometry g, int table, const double* __restrict__ widths, int n,
                                                      double* __restrict__ values, unsigned char* __restrict__ present) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const Row R{g, widths[r]};
  const double w = R.w, ext = g.ext, t_a = g.t_a, t_c = g.t_c;
  const int cols = table == 0 ? 8 : 3;
  double* out = values + (size_t)r * cols;
  unsigned char* has = present + (size_t)r * cols;
  for (int k = 0; k < cols; ++k) { out[k] = 0.0; has[k] = 0; }
  const double Ta = R.Tr(t_a), Tc = R.Tr(t_c);
  const Span cyl = cylinder(g, w);
  const Span at_beam = table == 0 ? sphere(g.c, g.omega_c, w, g.a + (t_a * g.omega_a)) : Span{p3(0, 0, 0), p3(0, 0, 0), false};
  const Span at_view = sphere(g.a, g.omega_a, w, g.c + (t_c * g.omega_c));
  auto nsd = [&](int k, double variance, double expectation) { out[k] = sqrt_f(variance) / expectation; has[k] = 1; };
  auto rrmse = [&](int k, double variance, double expectation, double reference) {
    out[k] = sqrt_f(variance + (sq(expectation - reference) / reference)); has[k] = 1;
  };
  // beam x point in 3D (BlP3D :184-216, rBlP3D :149-182)
  double blp_e = 0, blp_v = 0, blp_ref = 0;
  if (at_view.hit) {
    const double tam = len(g.a - at_view.entry), tap = len(g.a - at_view.exit);
    const double distance = len(g.c - ((at_view.entry + at_view.exit) / 2.0));
    const double tcm = fmax(t_c - w, distance - w), tcp = fmin(t_c + w, distance + w);
    const double A = R.TrP(tam, tap), C = R.TrP(tcm, tcp);
    blp_e = R.wpow(3) * A * C;
    blp_v = R.wpow(3 * 2) * ((sq(A) * (C / ext)) - (sq(A) * sq(C)));
    blp_ref = A * C;
  }
  if (table == 1) {
    if (cyl.hit) {
      double tcm = len(cyl.entry - g.c), tcp = len(cyl.exit - g.c);
      if (tcm == tcp) { tcm = tcm - w; tcp = tcp + w; }                                          // :403-406
      const double C = R.TrP(tcm, tcp);
      rrmse(0, R.wpow(2 * 2) * (R.blurred(Ta, tcm, tcp) - (sq(Ta) * sq(C))), R.wpow(2) * Ta * C, Ta * C);   // rBsBs2D1 :395-426
      rrmse(1, R.wpow(1 * 2) * ((Ta * sq(Tc)) - (sq(Ta) * sq(Tc))), R.wpow(1) * Ta * Tc, Ta * Tc);   // rBsBl1D :125-147
    }
    if (at_view.hit) rrmse(2, blp_v, blp_e, blp_ref);
    return;
  }
  if (cyl.hit) {
    const double tcm = len(cyl.entry - g.c), tcp = len(cyl.exit - g.c);
    const double C = R.TrP(tcm, tcp);
    nsd(0, R.wpow(2 * 2) * (R.blurred(Ta, tcm, tcp) - (sq(Ta) * sq(C))), R.wpow(2) * Ta * C);          // BsBs2D1 :428-453
    const double e1 = R.wpow(1) * Ta * Tc;
    nsd(1, R.wpow(1 * 2) * ((Ta * Tc) - (sq(Ta) * sq(Tc))), e1);                                        // BsBs1D :358-378
    const double tam = t_a - w, tap = t_a + w;
    const double A = R.TrP(tam, tap);
    nsd(2, R.wpow(2 * 2) * (R.blurred(sq(Tc), tam, tap) - (sq(Tc) * sq(A))), R.wpow(2) * A * Tc);      // BsBl2D2 :327-356
    nsd(4, R.wpow(1 * 2) * ((Ta * sq(Tc)) - (sq(Ta) * sq(Tc))), e1);                                    // BsBl1D :240-260
    nsd(5, R.wpow(1 * 2) * ((sq(Ta) * Tc) - (sq(Ta) * sq(Tc))), e1);                                    // BlBs1D :218-238
  }
  if (at_beam.hit) {
    const double tcm = len(at_beam.entry - g.c), tcp = len(at_beam.exit - g.c);
    const double distance = len(g.a - ((at_beam.entry + at_beam.exit) / 2.0));
    const double tam = fmax(t_a - w, distance - w), tap = fmin(t_a + w, distance + w);
    const double A = R.TrP(tam, tap), C = R.TrP(tcm, tcp);
    nsd(3, R.wpow(3 * 2) * (((A / ext) * sq(C)) - (sq(A) * sq(C))), R.wpow(3) * A * C);                 // PBl3D :293-325
    nsd(7, R.wpow(2 * 2) * (((A / ext) * Tc) - (sq(A) * sq(Tc))), R.wpow(2) * A * Tc);                  // PBs2D :262-291
  }
  if (at_view.hit) nsd(6, blp_v, blp_e);
}

// findRotationMatrix (Intersections.h:499-513) in float on narrowed arguments, as Analysis.h:549 calls it
void rotation_to_y(const double axis[3], double R_inv[3][3]) {
  const float c[3] = {(float)axis[0], (float)axis[1], (float)axis[2]}, w[3] = {0.0f, 1.0f, 0.0f};
  const float v[3] = {c[1] * w[2] - w[1] * c[2], c[2] * w[0] - w[2] * c[0], c[0] * w[1] - w[0] * c[1]};
  const float cs = c[0] * w[0] + c[1] * w[1] + c[2] * w[2];
  float vx[3][3] = {{0, v[2], -v[1]}, {-v[2], 0, v[0]}, {v[1], -v[0], 0}};   // [col][row]
  float vx2[3][3];
  for (int col = 0; col < 3; ++col)
    for (int row = 0; row < 3; ++row) vx2[col][row] = vx[0][row] * vx[col][0] + vx[1][row] * vx[col][1] + vx[2][row] * vx[col][2];
  const float k = 1 / (1 + cs);
  for (int col = 0; col < 3; ++col)
    for (int row = 0; row < 3; ++row) R_inv[col][row] = (double)((((col == row) ? 1.0f : 0.0f) + vx[col][row]) + (vx2[col][row] * k));
}

void invert(const double m[3][3], double r[3][3]) {
  const double ood = 1.0 / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
                            m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
  r[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood; r[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
  r[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood; r[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
  r[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood; r[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
  r[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood; r[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
  r[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" int pb_run_analysis(pb_ctx* ctx, double extinction, int table, double* first_col, double* values,
                               uint8_t* present, int capacity_rows, int* n_rows) {
  if (!ctx || !first_col || !values || !present || !n_rows || capacity_rows < 0 || (table != 0 && table != 1) || !(extinction > 0)) {
    pb_set_error("pb_run_analysis: bad argument"); return PB_ERR_ARG;
  }
  Geometry g;
  g.ext = extinction; g.mfp = 1 / extinction;
  const P3 view_point = p3(0, 0, 0);
  const P3 closest = p3(view_point.x + (0.0 * g.mfp), view_point.y, view_point.z + (10 * g.mfp));
  const P3 beam_origin = p3(closest.x + (0.0 * g.mfp), closest.y - (5 * g.mfp), closest.z);
  g.c = view_point; g.omega_c = p3(0, 0, 1);
  g.a = beam_origin; g.omega_a = p3(0, 1, 0);
  g.t_c = std::sqrt(dot(view_point - closest, view_point - closest));
  g.t_a = std::sqrt(dot(beam_origin - closest, beam_origin - closest));
  {
    const P3 diff = (g.a + ((g.omega_a * 10.0) * g.mfp)) - g.a;
    const double S_inv = 1 / std::sqrt(dot(diff, diff));
    const double axis[3] = {diff.x * S_inv, diff.y * S_inv, diff.z * S_inv};
    rotation_to_y(axis, g.R_inv.m);
    invert(g.R_inv.m, g.R.m);
  }
  // the kernel widths, accumulated as RunAnalysis' loops do (:92, :109)
  const double maximum_width = std::fmin(g.t_a, g.t_c);
  std::vector<double> widths;
  if (table == 0) { const double delta = 0.02 * g.mfp; for (double w = delta; w < maximum_width; w = w + delta) widths.push_back(w); }
  else for (double w = 1.0e-6 * g.mfp; w < maximum_width; w = w * 1.05) widths.push_back(w);
  const int n = (int)widths.size();
  *n_rows = n;
  if (n > capacity_rows) { pb_set_error("pb_run_analysis: capacity_rows too small (n_rows holds the need)"); return PB_ERR_CAPACITY; }
  if (n == 0) return PB_OK;
  PB_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int cols = table == 0 ? 8 : 3;
  DevBuf buf;
  const size_t nn = (size_t)n;
  PB_HIP(hipMalloc(&buf.p, nn * sizeof(double) * (1 + cols) + nn * cols));
  double* d_w = (double*)buf.p; double* d_v = d_w + nn; unsigned char* d_p = (unsigned char*)(d_v + nn * cols);
  PB_HIP(hipMemcpyAsync(d_w, widths.data(), sizeof(double) * nn, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(analysis_kernel, dim3((n + 63) / 64), dim3(64), 0, s, g, table, d_w, n, d_v, d_p);
  PB_HIP(hipGetLastError());
  PB_HIP(hipMemcpyAsync(values, d_v, sizeof(double) * nn * cols, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(present, d_p, nn * cols, hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  for (int r = 0; r < n; ++r) first_col[r] = widths[r] / g.mfp;
  return PB_OK;
}
