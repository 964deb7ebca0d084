// TEST INFRASTRUCTURE — CPU restatement of the reference's estimator variance study (Source/Analysis.h):
// normalised standard deviation and relative RMSE of eight photon point/beam estimators for one fixed
// geometry (a view ray along +z, a beam along +y, closest approach 10 and 5 mean free paths away), as a
// function of the blur kernel width.  Pinned by the reference's committed NSD.csv and, where the reference
// tree exists, by running its own RunAnalysis (oracle/ref_harness.cpp).  Built into liboracle.so.
#include <cmath>
#include <cstdint>

#include "orc_math.h"

namespace {
using namespace orc;

struct dmat3 { double m[3][3]; };   // m[col][row]
inline dvec3 mul(const dmat3& m, dvec3 v) {
  return {m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z, m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
          m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z};
}
inline dmat3 dinverse(const dmat3& a) {
  const double(*m)[3] = a.m;
  double ood = 1.0 / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
                      m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
  dmat3 r;
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
inline dvec3 operator/(dvec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline dvec3 operator-(dvec3 a) { return {-a.x, -a.y, -a.z}; }

// Intersections.h:499-513, float, as Analysis.h:549 calls it (dvec3 arguments narrow to vec3)
mat3 rotation(vec3 c_unit_dir, vec3 w_unit_dir) {
  vec3 v = cross(c_unit_dir, w_unit_dir);
  float c = dot(c_unit_dir, w_unit_dir);
  mat3 v_x;
  v_x.m[0][0] = 0;    v_x.m[1][0] = -v.z; v_x.m[2][0] = v.y;
  v_x.m[0][1] = v.z;  v_x.m[1][1] = 0;    v_x.m[2][1] = -v.x;
  v_x.m[0][2] = -v.y; v_x.m[1][2] = v.x;  v_x.m[2][2] = 0;
  return identity3() + v_x + ((v_x * v_x) * (1 / (1 + c)));
}

struct Study {
  double ext, mfp;
  dvec3 a, c, omega_a, omega_c;
  double t_a, t_c;
  struct Hit { dvec3 entry{0, 0, 0}, exit{0, 0, 0}; };

  double Tr(double len) const { return std::exp(-ext * len); }                       // Analysis.h:477-480
  double TrPrime(double t_min, double t_plus) const { return (Tr(t_min) - Tr(t_plus)) / ext; }   // :482-485

  void SetUp() {                                                                      // :455-475
    dvec3 view_point{0, 0, 0};
    dvec3 closest{view_point.x + (0.0 * mfp), view_point.y, view_point.z + (10 * mfp)};
    dvec3 beam_origin{closest.x + (0.0 * mfp), closest.y - (5 * mfp), closest.z};
    c = view_point; omega_c = dvec3{0, 0, 1};
    a = beam_origin; omega_a = dvec3{0, 1, 0};
    t_c = dlength(view_point - closest);
    t_a = dlength(beam_origin - closest);
  }

  bool Sphere(dvec3 p, dvec3 dir, double kw, dvec3 centre, Hit& h) const {           // :487-528
    dvec3 vpc = centre - p;
    double cos_theta = ddot(vpc, dir) / (dlength(vpc) * dlength(dir));
    double distance = cos_theta * dlength(vpc);
    dvec3 pc = p + (distance * dir);
    if (ddot(vpc, dir) < 0) {
      if (dlength(vpc) > kw) return false;
      else if (dlength(vpc) == kw) { h.entry = p; h.exit = p + ((2 * kw) * dir); }
      else {
        double dist = std::sqrt(std::pow(kw, 2) - std::pow(dlength(pc - centre), 2));
        double di1 = dist - dlength(pc - p);
        h.exit = p + (dir * di1);
        h.entry = p + (dir * di1) - (dir * (kw * 2));
      }
    } else {
      if (dlength(centre - pc) > kw) return false;
      double dist = std::sqrt(std::pow(kw, 2) - std::pow(dlength(pc - centre), 2));
      double di1 = (dlength(vpc) > kw) ? dlength(pc - p) - dist : dlength(pc - p) + dist;
      h.entry = p + (dir * di1);
      double di2 = di1 + (2 * dlength(pc - h.entry));
      h.exit = p + (dir * di2);
    }
    return true;
  }

  bool Cylinder(double kw, Hit& h) const {                                           // :530-653
    bool first_inte = false;
    dvec3 beam_end = a + ((omega_a * 10.0) * mfp);
    dvec3 difference = beam_end - a;
    double len = dlength(difference);
    dvec3 w_unit = dnormalize(dvec3{0.0, len, 0.0});
    double S_inv = 1 / len;
    dvec3 c_unit = difference * S_inv;
    mat3 Rf = rotation(V3((float)c_unit.x, (float)c_unit.y, (float)c_unit.z), V3((float)w_unit.x, (float)w_unit.y, (float)w_unit.z));
    dmat3 R_inv;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R_inv.m[i][j] = (double)Rf.m[i][j];
    dvec3 T_inv = -(mul(R_inv, a) * S_inv);
    dvec3 o = mul(R_inv, c * S_inv) + T_inv;
    dvec3 d = mul(R_inv, omega_c);
    dvec3 first{0, 0, 0}, second{0, 0, 0};
    double aq = d.x * d.x + d.z * d.z;
    double bq = 2 * o.x * d.x + 2 * o.z * d.z;
    double cq = o.x * o.x + o.z * o.z - std::pow(kw * S_inv, 2);
    double b24ac = bq * bq - 4 * aq * cq;
    if (b24ac < 0) return false;
    double sq = sqrtf((float)b24ac);                                                  // :577 sqrtf on a double
    double t0 = (-bq + sq) / (2 * aq), t1 = (-bq - sq) / (2 * aq);
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    double y0 = o.y + t0 * d.y, y1 = o.y + t1 * d.y;
    if (y0 < 0) {
      if (y1 < 0) return false;
      double th1 = t0 + (t1 - t0) * y0 / (y0 - y1);
      if (th1 <= 0) return false;
      first = o + (d * th1); first_inte = true;
      if (y1 <= 1) second = o + (d * t1);
      else { double th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) second = o + (d * th2); }
    } else if (y0 >= 0 && y0 <= 1) {
      if (t0 <= 0) return false;
      first = o + (d * t0); first_inte = true;
      if (y1 >= 0) {
        if (y1 <= 1) second = o + (d * t1);
        else { double th2 = t1 + (t1 - t0) * (y1 - 1) / (y0 - y1); if (th2 > 0) second = o + (d * th2); }
      } else { double th2 = t0 + (t1 - t0) * (y0) / (y0 - y1); if (th2 > 0) second = o + (d * th2); }
    } else if (y0 > 1) {
      if (y1 > 1) return false;
      double th = t0 + (t1 - t0) * (y0 - 1) / (y0 - y1);
      if (th <= 0) return false;
      first = o + (d * th); first_inte = true;
      if (y1 >= 0) second = o + (d * t1);
      else { double th2 = t0 + (t1 - t0) * (y0) / (y0 - y1); if (th2 > 0) second = o + (d * th2); }
    }
    dmat3 R = dinverse(R_inv);
    h.entry = mul(R, first - T_inv) / S_inv;
    h.exit = mul(R, second - T_inv) / S_inv;
    return first_inte;
  }

  // variance of a short-beam estimate blurred along the other path (:344-348, :441-445)
  double blur_var(double w, int dim, double tr_other, double t_minus, double t_plus) const {
    return std::pow(w, -dim * 2) * (tr_other * (2 / std::pow(ext, 2)) * (Tr(t_minus) + (Tr(t_plus) * (((t_minus - t_plus) * ext) - 1))));
  }
  static double nsd_of(double variance, double expectation) { return sqrtf((float)variance) / expectation; }

  // The eight NSD columns in RunAnalysis' order (:94-101); false = empty cell.
  bool nsd_column(int col, double w, double& out) const {
    Hit h;
    switch (col) {
      case 0: {                                                                      // BsBs2D1 :428-453
        if (!Cylinder(w, h)) return false;
        double tcm = dlength(h.entry - c), tcp = dlength(h.exit - c);
        double e = std::pow(w, -2) * Tr(t_a) * TrPrime(tcm, tcp);
        double v = std::pow(w, -2 * 2) * (Tr(t_a) * (2 / std::pow(ext, 2)) * (Tr(tcm) + (Tr(tcp) * (((tcm - tcp) * ext) - 1))) -
                                          (std::pow(Tr(t_a), 2) * std::pow(TrPrime(tcm, tcp), 2)));
        out = nsd_of(v, e); return true;
      }
      case 1: {                                                                      // BsBs1D :358-378
        if (!Cylinder(w, h)) return false;
        double e = std::pow(w, -1) * Tr(t_a) * Tr(t_c);
        double v = std::pow(w, -1 * 2) * ((Tr(t_a) * Tr(t_c)) - (std::pow(Tr(t_a), 2) * std::pow(Tr(t_c), 2)));
        out = nsd_of(v, e); return true;
      }
      case 2: {                                                                      // BsBl2D2 :327-356
        if (!Cylinder(w, h)) return false;
        double tam = t_a - w, tap = t_a + w;
        double e = std::pow(w, -2) * TrPrime(tam, tap) * Tr(t_c);
        double v = std::pow(w, -2 * 2) * (std::pow(Tr(t_c), 2) * (2 / std::pow(ext, 2)) * (Tr(tam) + (Tr(tap) * (((tam - tap) * ext) - 1))) -
                                          (std::pow(Tr(t_c), 2) * std::pow(TrPrime(tam, tap), 2)));
        out = nsd_of(v, e); return true;
      }
      case 3: {                                                                      // PBl3D :293-325
        dvec3 centre = a + (t_a * omega_a);
        if (!Sphere(c, omega_c, w, centre, h)) return false;
        double tcm = dlength(h.entry - c), tcp = dlength(h.exit - c);
        dvec3 mid = (h.entry + h.exit) / 2.0;
        double distance = dlength(a - mid);
        double tam = std::fmax(t_a - w, distance - w), tap = std::fmin(t_a + w, distance + w);
        double e = std::pow(w, -3) * TrPrime(tam, tap) * TrPrime(tcm, tcp);
        double v = std::pow(w, -3 * 2) * (((TrPrime(tam, tap) / ext) * std::pow(TrPrime(tcm, tcp), 2)) -
                                          (std::pow(TrPrime(tam, tap), 2) * std::pow(TrPrime(tcm, tcp), 2)));
        out = nsd_of(v, e); return true;
      }
      case 4: {                                                                      // BsBl1D :240-260
        if (!Cylinder(w, h)) return false;
        double e = std::pow(w, -1) * Tr(t_a) * Tr(t_c);
        double v = std::pow(w, -1 * 2) * ((Tr(t_a) * std::pow(Tr(t_c), 2)) - (std::pow(Tr(t_a), 2) * std::pow(Tr(t_c), 2)));
        out = nsd_of(v, e); return true;
      }
      case 5: {                                                                      // BlBs1D :218-238
        if (!Cylinder(w, h)) return false;
        double e = std::pow(w, -1) * Tr(t_a) * Tr(t_c);
        double v = std::pow(w, -1 * 2) * ((std::pow(Tr(t_a), 2) * Tr(t_c)) - (std::pow(Tr(t_a), 2) * std::pow(Tr(t_c), 2)));
        out = nsd_of(v, e); return true;
      }
      case 6: {                                                                      // BlP3D :184-216
        dvec3 point = c + (t_c * omega_c);
        if (!Sphere(a, omega_a, w, point, h)) return false;
        double tam = dlength(a - h.entry), tap = dlength(a - h.exit);
        dvec3 mid = (h.entry + h.exit) / 2.0;
        double distance = dlength(c - mid);
        double tcm = std::fmax(t_c - w, distance - w), tcp = std::fmin(t_c + w, distance + w);
        double e = std::pow(w, -3) * TrPrime(tam, tap) * TrPrime(tcm, tcp);
        double v = std::pow(w, -3 * 2) * ((std::pow(TrPrime(tam, tap), 2) * (TrPrime(tcm, tcp) / ext)) -
                                          (std::pow(TrPrime(tam, tap), 2) * std::pow(TrPrime(tcm, tcp), 2)));
        out = nsd_of(v, e); return true;
      }
      default: {                                                                     // PBs2D :262-291
        dvec3 centre = a + (t_a * omega_a);
        if (!Sphere(c, omega_c, w, centre, h)) return false;
        dvec3 mid = (h.entry + h.exit) / 2.0;
        double distance = dlength(a - mid);
        double tam = std::fmax(t_a - w, distance - w), tap = std::fmin(t_a + w, distance + w);
        double e = std::pow(w, -2) * TrPrime(tam, tap) * Tr(t_c);
        double v = std::pow(w, -2 * 2) * (((TrPrime(tam, tap) / ext) * Tr(t_c)) - (std::pow(TrPrime(tam, tap), 2) * std::pow(Tr(t_c), 2)));
        out = nsd_of(v, e); return true;
      }
    }
  }

  // The three rRMSE columns (:111-117): rBsBs2D1, rBsBl1D, rBlP3D.
  bool rrmse_column(int col, double w, double& out) const {
    Hit h;
    double e, v, reference;
    if (col == 0) {                                                                  // :395-426
      if (!Cylinder(w, h)) return false;
      double tcm = dlength(h.entry - c), tcp = dlength(h.exit - c);
      if (tcm == tcp) { tcm = tcm - w; tcp = tcp + w; }
      e = std::pow(w, -2) * Tr(t_a) * TrPrime(tcm, tcp);
      v = std::pow(w, -2 * 2) * (Tr(t_a) * (2 / std::pow(ext, 2)) * (Tr(tcm) + (Tr(tcp) * (((tcm - tcp) * ext) - 1))) -
                                 (std::pow(Tr(t_a), 2) * std::pow(TrPrime(tcm, tcp), 2)));
      reference = Tr(t_a) * TrPrime(tcm, tcp);
    } else if (col == 1) {                                                           // :125-147
      if (!Cylinder(w, h)) return false;
      e = std::pow(w, -1) * Tr(t_a) * Tr(t_c);
      v = std::pow(w, -1 * 2) * ((Tr(t_a) * std::pow(Tr(t_c), 2)) - (std::pow(Tr(t_a), 2) * std::pow(Tr(t_c), 2)));
      reference = Tr(t_a) * Tr(t_c);
    } else {                                                                         // :149-182
      dvec3 point = c + (t_c * omega_c);
      if (!Sphere(a, omega_a, w, point, h)) return false;
      double tam = dlength(a - h.entry), tap = dlength(a - h.exit);
      dvec3 mid = (h.entry + h.exit) / 2.0;
      double distance = dlength(c - mid);
      double tcm = std::fmax(t_c - w, distance - w), tcp = std::fmin(t_c + w, distance + w);
      e = std::pow(w, -3) * TrPrime(tam, tap) * TrPrime(tcm, tcp);
      v = std::pow(w, -3 * 2) * ((std::pow(TrPrime(tam, tap), 2) * (TrPrime(tcm, tcp) / ext)) -
                                 (std::pow(TrPrime(tam, tap), 2) * std::pow(TrPrime(tcm, tcp), 2)));
      reference = TrPrime(tam, tap) * TrPrime(tcm, tcp);
    }
    double sqr_bias = std::pow(e - reference, 2);
    out = sqrtf((float)(v + (sqr_bias / reference)));
    return true;
  }
};
}  // namespace

extern "C" {

// RunAnalysis (Analysis.h:81-122).  table 0 = NSD.csv (8 columns), 1 = rRMSE.csv (3 columns).  Row r:
// first_col[r] = kernel_width / mfp, values[r * cols + k], present[r * cols + k] = 0 for an empty cell.
// Returns the number of rows of the table (fills at most `capacity`).
int orc_run_analysis(double extinction, int table, double* first_col, double* values, uint8_t* present, int capacity) {
  Study s;
  s.ext = extinction; s.mfp = 1 / extinction;
  s.SetUp();
  const double maximum_width = std::fmin(s.t_a, s.t_c);
  const int cols = table == 0 ? 8 : 3;
  int rows = 0;
  auto row = [&](double w) {
    if (rows < capacity) {
      first_col[rows] = w / s.mfp;
      for (int k = 0; k < cols; ++k) {
        double v = 0.0;
        bool ok = table == 0 ? s.nsd_column(k, w, v) : s.rrmse_column(k, w, v);
        values[rows * cols + k] = ok ? v : 0.0;
        present[rows * cols + k] = ok ? 1 : 0;
      }
    }
    rows++;
  };
  if (table == 0) {
    const double delta = 0.02 * s.mfp;
    for (double w = delta; w < maximum_width; w = w + delta) row(w);
  } else {
    for (double w = 1.0e-6 * s.mfp; w < maximum_width; w = w * 1.05) row(w);
  }
  return rows;
}

}  // extern "C"
