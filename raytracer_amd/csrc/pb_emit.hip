// CastPhotonBeams / CastBeam / FindDirection (CastPhotons.h:312-633, :1128-1141)
// and BoundPhotonBeams (CastPhotons.h:144-288) on the GPU.
//
// One lane per photon.  The reference draws from one sequential std::mt19937;
// here draw k of photon p is philox4x32-10(counter = (p, k, 0, 0), key = seed),
// consumed in the reference's order (its recursion is walked depth-first with
// an explicit stack), so the photon streams are independent and the result does
// not depend on the launch geometry.  Beams and segments are written at offsets
// obtained from prefix sums of per-photon / per-beam counts, i.e. in the order
// the reference's sequential push_back would give: every kernel runs twice,
// first counting, then writing.
#include <cstddef>
#include <cstring>

#include <utility>

#include <cstdlib>

#include "pb_internal.h"
#include "pb_philox.h"
#include "pb_geom.h"
#include "pb_grid.h"
#include "pb_scan.h"

namespace {

constexpr int EMIT_THREADS = 64;
constexpr int MAX_STACK = 8;   // nesting of in-medium scattering events kept per photon

struct Rng {
  uint32_t photon, draw, k0, k1;
  PB_DI double canonical() {
    return philox_canonical(photon, draw++, k0, k1);
  }
  PB_DI double range(double a, double b) { return canonical() * (b - a) + a; }
  PB_DI double uniform() { return range(0.0, 1.0); }                           // CastPhotons.h:87
  PB_DI double uniform_offset() { return range(-0.01, 0.01); }                 // :88
  PB_DI double uniform_radius() { return range(0.0, 0.05); }                   // :89
  PB_DI double uniform_beam() { return range((double)0.02f, (double)0.03f); }  // :90
  PB_DI double uniform_small() { return range((double)0.02f, (double)0.06f); } // :91
};

struct EmitParams {
  const TriPrep* tris;
  int n_tri;
  float matrix[16];
  pb_control ctl;
  uint64_t seed;
  int n_photons;
  uint32_t* cnt_const;        // counting pass: pushes of each photon into the two lists
  uint32_t* cnt_scatt;
  const uint32_t* off_const;  // writing pass: where the photon's pushes start
  const uint32_t* off_scatt;
  pb_beam* beams_const;
  pb_beam* beams_scatt;
  unsigned int* dropped;      // scattering events dropped because the stack was full
  GridView grid;              // extinction field; used when ctl.heterogeneous
  const pb_beam* sources;     // re-cast pass: the beams to cast again (one per lane instead of a photon)
  int staged;                 // > 0: writing pass without a counting pass before it: every photon writes into `staged` slots per
                              // list of its own in beams_const (= the staging buffer) and leaves its counts; gather_staged_kernel
                              // moves the beams to their places once the offsets are known
  unsigned int* overflow;     // staged pass: set when a photon pushed more beams into a list than it has slots for
};
// Slots per photon and list of the staged pass: what a photon can push grows with its bounces (one beam per bounce, and
// one more per scattering event on the way, into the other list), so the slots are sized from the bounce budget - 80
// bytes a slot, two lists: 640 B a photon without bounces, 2560 B at most.  A photon that outgrows them sends the whole
// pass the two-pass way, so the count only decides how often that happens.
constexpr int EMIT_STAGE_MAX = 16;
PB_HDI int emit_stage_slots(int bounces) { const int want = 2 * (bounces + 1) + 2; return want < EMIT_STAGE_MAX ? want : EMIT_STAGE_MAX; }
constexpr int EMIT_STAGE_MAX_PHOTONS = 1 << 18;  // 2 x 16 x 80 B per photon: 655 MB at most

// CastPhotons.h:1128-1141
PB_DI V4 find_direction(Rng& rng, V4 origin, V4 centre) {
  const float r1 = (float)rng.uniform();
  const float r2 = (float)rng.uniform();
  const float the = (float)(2 * PB_PI_REF * r1);
  const float phi = sqrtf(r2);
  const float x = centre.x + (phi * cosf(the));
  const float z = centre.z + (phi * sinf(the));
  const float y = centre.y + sqrtf(1 - r2);
  return v4(x, y, z, 1.0f) - origin;
}

// CastPhotons.h:635-647
PB_DI bool intersect_plane(V3 n, V3 p0, V3 l0, V3 l, float& t) {
  const float denom = dot3(n, l);
  if (fabsf(denom) > 1e-6) {
    const V3 p0l0 = p0 - l0;
    t = dot3(p0l0, n) / denom;
    return t >= 0;
  }
  return false;
}

// CastPhotons.h:1143-1146
PB_DI double transmittance(double length, double ext) { return exp(-ext * length); }

struct Frame {
  V4 origin, direction, hit_pos;
  V3 energy;
  float t_s, t_end, scattered, avg_ext;
  int bounce, parity, stage, hit_index;
  bool hit, scatt, ada;
};

struct BeamSink {
  const EmitParams* p;
  uint32_t photon, order, n_const, n_scatt, off_const, off_scatt;
  uint32_t cap;               // beams a list has room for at its offset (staged pass: its slots; else no limit)
  bool scattered_only;        // re-cast pass: pushes into the first list are counted in `order` but not kept
};

template <bool WRITE>
PB_DI void push_beam(BeamSink& sink, int parity, V4 start, V4 end, V4 omega_u, float radius, V3 energy,
                     int index_hit, bool ada, bool absorbed) {
  if (sink.scattered_only && !parity) { sink.order++; return; }
  if (WRITE) {
    pb_beam b;
    b.start[0] = start.x; b.start[1] = start.y; b.start[2] = start.z; b.start[3] = start.w;
    b.end[0] = end.x; b.end[1] = end.y; b.end[2] = end.z; b.end[3] = end.w;
    b.omega_u[0] = omega_u.x; b.omega_u[1] = omega_u.y; b.omega_u[2] = omega_u.z; b.omega_u[3] = omega_u.w;
    b.radius = radius;
    b.energy[0] = energy.x; b.energy[1] = energy.y; b.energy[2] = energy.z;
    b.index_hit = index_hit;
    b.flags = (ada ? 1u : 0u) | (absorbed ? 4u : 0u) | (parity ? 8u : 0u);   // scattered is never set (CastPhotons.h:398)
    b.photon = sink.photon;
    b.order = sink.order;
    if (parity) { if (sink.n_scatt < sink.cap) sink.p->beams_scatt[sink.off_scatt + sink.n_scatt] = b; }
    else if (sink.n_const < sink.cap) sink.p->beams_const[sink.off_const + sink.n_const] = b;
  }
  sink.order++;
  if (parity) sink.n_scatt++; else sink.n_const++;
}

// The recursion of CastBeam (CastPhotons.h:392-633) as an explicit depth-first
// walk: a frame is visited at stage 0 (entry up to the optional scattered
// child) and at stage 1 (push of its own beam, then the bounce as a tail call).
template <bool WRITE>
PB_DI void cast_beam(const EmitParams& p, Rng& rng, BeamSink& sink, Frame first, V4 omega_u, float radius) {
  const pb_control& C = p.ctl;
  const float extinction_c = C.absorption_c + C.scattering_c;   // Control.h:48
  Frame st[MAX_STACK];
  int sp = 0;
  st[0] = first;
  while (sp >= 0) {
    Frame& F = st[sp];
    if (F.stage == 0) {
      V4 pos = v4(0, 0, 0, 0);
      float dist; int idx;
      const bool hit = closest_intersection(p.tris, p.n_tri, F.origin, F.direction, pos, dist, idx);
      float t_s;
      if (hit) {
        t_s = length3(xyz(pos) - xyz(F.origin));
        if (t_s < 0.0005) { sp--; continue; }
      } else t_s = PB_ACTUAL_WIDTH;
      // CastPhotons.h:409-415: mean extinction along the way to the surface
      const float t_end = t_s;
      const V3 walk_a = xyz(F.origin), walk_b = xyz(F.origin + (F.direction * t_end));
      const float average_extinction = C.heterogeneous ? extinction3d_mean(p.grid, walk_a, walk_b) : extinction_c;
      const float scattered = (float)rng.uniform();
      bool scatt = false, spawn = false;
      Frame child;
      if ((scattered <= (C.scattering_c / extinction_c)) && C.scatter) {
        const float tmp_t_s = (float)(-(log(1 - rng.uniform()) / average_extinction));
        if ((tmp_t_s < t_s) && (tmp_t_s > 0)) {
          t_s = tmp_t_s;
          V4 start = F.origin + (t_s * normalize4(F.direction));
          start.w = 1;
          const float the = (float)(2 * PB_PI_REF * rng.uniform());
          const float phi = (float)acos(1 - 2 * rng.uniform());
          const float x = sinf(phi) * cosf(the);
          const float y = sinf(phi) * sinf(the);
          const float z = cosf(phi);
          const V3 dir = normalize3(v3(x, y, z));
          // CastPhotons.h:436-447
          const float transmitted = (float)transmittance(
              t_s, C.heterogeneous ? extinction3d_mean_until(p.grid, walk_a, walk_b, t_s) : average_extinction);
          V3 ne = F.energy * transmitted;
          if (!(ne.x == 0 && ne.y == 0 && ne.z == 0)) {
            ne.x = fminf(F.energy.x, fmaxf(ne.x, 0.f));
            ne.y = fminf(F.energy.y, fmaxf(ne.y, 0.f));
            ne.z = fminf(F.energy.z, fmaxf(ne.z, 0.f));
            child.origin = start;
            child.direction = v4(dir.x, dir.y, dir.z, 1.0f);
            child.energy = ne;
            child.bounce = F.bounce;
            child.parity = F.parity ^ 1;   // add_to and other swap (CastPhotons.h:458-460)
            child.stage = 0;
            child.ada = false;
            spawn = true;
          }
          if (C.short_beams) scatt = true;
        }
      } else {
        const float tmp_t_s = (float)(-(log(1 - rng.uniform()) / average_extinction));
        if ((tmp_t_s < t_s) && (tmp_t_s > 0)) { sp--; continue; }   // absorbed in flight, never pushed (:469-479)
      }
      F.t_s = t_s; F.t_end = t_end; F.scattered = scattered; F.avg_ext = average_extinction;
      F.hit = hit; F.hit_pos = pos; F.hit_index = idx; F.scatt = scatt;
      F.stage = 1;
      if (spawn) {
        if (sp + 1 < MAX_STACK) { st[++sp] = child; continue; }
        atomicAdd(p.dropped, 1u);
      }
    }
    // stage 1: the second ClosestIntersection of the reference has the same
    // arguments as the first, its result is the stored one
    const V4 origin = F.origin, direction = F.direction;
    const V3 energy = F.energy;
    if (F.hit) {
      V4 end;
      if (F.scatt) { end = origin + (F.t_s * direction); end.w = 1.0f; }
      else end = F.hit_pos;
      const float absorbed = (float)rng.uniform();
      const bool is_absorbed = absorbed < C.absorbed;
      push_beam<WRITE>(sink, F.parity, origin, end, omega_u, radius, energy, F.hit_index, F.ada, is_absorbed);
      if ((F.bounce < C.bounces) && (!is_absorbed) && (!F.scatt)) {
        const TriPrep& T = p.tris[F.hit_index];
        const V4 normal = v4(T.n[0], T.n[1], T.n[2], 1.0f);
        const V4 incident = F.hit_pos + direction;
        const V3 refl = reflect3(xyz(incident), xyz(normal));
        const V3 rn = normalize3(refl);
        const V4 reflected = v4(rn.x, rn.y, rn.z, 1.0f);
        const V4 bounce_dir = find_direction(rng, F.hit_pos, F.hit_pos + (reflected * 0.5f));
        const V3 bn = normalize3(xyz(bounce_dir));
        V4 new_direction = v4(bn.x, bn.y, bn.z, 1.0f);
        if (F.scattered > 0.8) new_direction = reflected;
        // CastPhotons.h:529-540: the sample list is still the one of the walk to the surface
        float ext_to_ts = F.avg_ext;
        if (C.heterogeneous)
          ext_to_ts = extinction3d_mean_until(p.grid, xyz(origin), xyz(origin + (direction * F.t_end)), F.t_s);
        const float transmitted = (float)transmittance(F.t_s, ext_to_ts);
        V3 ne = energy * transmitted;
        if (!(ne.x == 0 && ne.y == 0 && ne.z == 0)) {
          ne.x = fminf(energy.x, fmaxf(ne.x, 0.f));
          ne.y = fminf(energy.y, fmaxf(ne.y, 0.f));
          ne.z = fminf(energy.z, fmaxf(ne.z, 0.f));
          Frame b;
          b.origin = F.hit_pos + (normal * 0.0001f);
          b.direction = new_direction;
          b.energy = ne;
          b.bounce = F.bounce + 1;
          b.parity = F.parity ^ 1;
          b.stage = 0;
          b.ada = false;
          st[sp] = b;   // tail call: nothing follows the bounce in the parent
          continue;
        }
      }
      sp--;
    } else {
      // the beam leaves through the open front: clip at four planes through the camera (:558-618)
      const float* M = p.matrix;
      const V4 top_left = v4(-1, -1, -1, 1), top_right = v4(1, -1, -1, 1);
      const V4 bot_left = v4(-1, 1, -1, 1), bot_right = v4(1, 1, -1, 1);
      const V4 converge = mul4(M, v4(0, 0, C.camera[2], 1.0f));
      const V4 top = mul4(M, (top_left + top_right) / 2.0f);
      const V4 right = mul4(M, (top_right + bot_right) / 2.0f);
      const V4 bot = mul4(M, (bot_right + bot_left) / 2.0f);
      const V4 left = mul4(M, (bot_left + top_left) / 2.0f);
      const V3 p00 = xyz((top + converge) / 2.0f), p01 = xyz((right + converge) / 2.0f);
      const V3 p02 = xyz((bot + converge) / 2.0f), p03 = xyz((left + converge) / 2.0f);
      const V3 n00 = normalize3(cross3(p00 - xyz(top), xyz(mul4(M, top_left) - top)));
      const V3 n01 = normalize3(cross3(p01 - xyz(right), xyz(mul4(M, top_right) - right)));
      const V3 n02 = normalize3(cross3(p02 - xyz(bot), xyz(mul4(M, bot_right) - bot)));
      const V3 n03 = normalize3(cross3(p03 - xyz(left), xyz(mul4(M, bot_left) - left)));
      const V3 dir = normalize3(xyz(direction));
      const V3 start = xyz(origin);
      float t, t_min = 3.402823466e+38f;
      if (intersect_plane(n00, p00, start, dir, t)) if (t < t_min) t_min = t;
      if (intersect_plane(n01, p01, start, dir, t)) if (t < t_min) t_min = t;
      if (intersect_plane(n02, p02, start, dir, t)) if (t < t_min) t_min = t;
      if (intersect_plane(n03, p03, start, dir, t)) if (t < t_min) t_min = t;
      if (t_min != 3.402823466e+38f) {
        V4 end;
        if (F.scatt) { end = origin + (F.t_s * direction); end.w = 1.0f; }
        else { const V3 e = start + (t_min * dir); end = v4(e.x, e.y, e.z, 1.0f); }
        push_beam<WRITE>(sink, F.parity, origin, end, omega_u, radius, energy, -1, F.ada, false);
      }
      sp--;
    }
  }
}

// CastPhotonBeams, CastPhotons.h:312-368: one photon per lane.  RECAST: RecastPhotonBeams (:290-310), one
// existing beam per lane cast again from its start along its direction; the reference loops over the vector
// it appends to and never ends, here exactly the beams present on entry are re-cast.  As there, only what
// lands in the scattered list is kept.
template <bool WRITE, bool RECAST>
__global__ __launch_bounds__(EMIT_THREADS) void emit_kernel(const EmitParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_photons) return;
  const pb_control& C = p.ctl;
  Rng rng;
  rng.photon = (uint32_t)i; rng.draw = 0;
  rng.k0 = (uint32_t)p.seed; rng.k1 = (uint32_t)(p.seed >> 32);
  BeamSink sink;
  sink.p = &p; sink.photon = (uint32_t)i; sink.order = 0; sink.n_const = 0; sink.n_scatt = 0;
  const bool staged = WRITE && p.staged != 0;
  const uint32_t slots = (uint32_t)p.staged;
  sink.cap = staged ? slots : 0xffffffffu;
  if (staged) { sink.off_const = (uint32_t)i * 2u * slots; sink.off_scatt = sink.off_const + slots; }
  else {
    sink.off_const = (WRITE && !RECAST) ? p.off_const[i] : 0u;
    sink.off_scatt = WRITE ? p.off_scatt[i] : 0u;
  }
  sink.scattered_only = RECAST;
  if (RECAST) {
    const pb_beam b = p.sources[i];
    const V3 d = normalize3(v3(b.end[0] - b.start[0], b.end[1] - b.start[1], b.end[2] - b.start[2]));
    Frame f;
    f.origin = v4(b.start[0], b.start[1], b.start[2], b.start[3]);
    f.direction = v4(d.x, d.y, d.z, 1.0f);
    f.energy = v3(b.energy[0], b.energy[1], b.energy[2]);
    f.bounce = 0; f.parity = 0; f.stage = 0; f.ada = (b.flags & 1u) != 0;
    cast_beam<WRITE>(p, rng, sink, f, v4(b.omega_u[0], b.omega_u[1], b.omega_u[2], b.omega_u[3]), b.radius);
    if (!WRITE || staged) { p.cnt_const[i] = 0u; p.cnt_scatt[i] = sink.n_scatt; }
    if (staged && sink.n_scatt > slots) atomicOr(p.overflow, 1u);
    return;
  }

  const V4 light = v4(C.light_position[0], C.light_position[1], C.light_position[2], C.light_position[3]);
  const V4 origin = mul4(p.matrix, light);
  const V4 centre = v4(origin.x, (float)(origin.y + 1.5), origin.z, 1.0f);
  V4 direction = find_direction(rng, origin, centre);
  direction = normalize4(direction);
  float r = (float)rng.uniform_beam();
  float r_small = (float)rng.uniform_small();
  if (C.fixed_radius) { r = C.radius; r_small = r; }
  const V3 w_u = normalize3(v3(direction.x + r_small, direction.y, direction.z));
  const V4 omega_u = v4(w_u.x, w_u.y, w_u.z, 1.0f);
  const float t = (float)(2 * PB_PI_REF * rng.uniform_radius());
  const double u1 = rng.uniform_radius();
  const double u2 = rng.uniform_radius();
  const float u = (float)(u1 + u2);
  const float w_width = (u > 1) ? 2 - u : u;
  const float x_offset = w_width * cosf(t);
  const float y_offset = (float)(rng.uniform_offset() * 0.1);
  const float z_offset = w_width * sinf(t);
  Frame f;
  f.origin = origin + v4(x_offset, y_offset, z_offset, 0);
  f.direction = direction;
  f.energy = v3(C.light_power[0], C.light_power[1], C.light_power[2]);
  f.bounce = 0; f.parity = 0; f.stage = 0; f.ada = C.adaptive != 0;
  cast_beam<WRITE>(p, rng, sink, f, omega_u, r);
  if (!WRITE || staged) { p.cnt_const[i] = sink.n_const; p.cnt_scatt[i] = sink.n_scatt; }
  if (staged && (sink.n_const > slots || sink.n_scatt > slots)) atomicOr(p.overflow, 1u);
}

// The beams of a staged pass to their places in the two lists: one lane per photon (or re-cast beam).
__global__ __launch_bounds__(256) void gather_staged_kernel(const pb_beam* __restrict__ stage, const uint32_t* __restrict__ cnt_const,
                                                           const uint32_t* __restrict__ cnt_scatt, const uint32_t* __restrict__ off_const,
                                                           const uint32_t* __restrict__ off_scatt, pb_beam* __restrict__ beams_const,
                                                           pb_beam* __restrict__ beams_scatt, int n, int slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pb_beam* mine = stage + (size_t)i * 2 * slots;
  const uint32_t nc = cnt_const[i], ns = cnt_scatt[i];
  for (uint32_t k = 0; k < nc; ++k) beams_const[off_const[i] + k] = mine[k];
  for (uint32_t k = 0; k < ns; ++k) beams_scatt[off_scatt[i] + k] = mine[slots + k];
}

// ---------------------------------------------------------------- segments --
struct BoundParams {
  const pb_beam* beams;       // beams_const list then beams_scattered list
  int n_const, n_scatt;
  float extinction_c;
  int heterogeneous;
  GridView grid;
  uint32_t* seg_count;        // counting pass
  const uint32_t* seg_offset; // writing pass
  pb_segment* segs;
};

// BoundPhotonBeams (CastPhotons.h:144-288).  Homogeneous medium: the whole beam
// becomes one segment, plus a second, ~1e-7-long one whenever the float length of
// prior + dir*len - start falls short of len (:196-202).  Heterogeneous medium: one
// segment per interval between consecutive samples of Extinction3D along the beam,
// carrying the extinction at both ends and the running mean up to it (:176-194, :270-277).
template <bool WRITE>
__global__ __launch_bounds__(256) void bound_kernel(const BoundParams p) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = p.n_const + p.n_scatt;
  if (g >= n) return;
  const pb_beam b = p.beams[g];
  const int i = (g < p.n_const) ? g : g - p.n_const;   // index in its own list: PhotonSeg.id (:227)
  const V4 start = v4(b.start[0], b.start[1], b.start[2], b.start[3]);
  const V4 end = v4(b.end[0], b.end[1], b.end[2], b.end[3]);
  const V3 diff = xyz(end) - xyz(start);
  const V3 dir = normalize3(diff);
  const float len = length3(diff);
  float j = 0;
  const V3 x_dir = normalize3(v3(1, 0, 0)), y_dir = normalize3(v3(0, 1, 0)), z_dir = normalize3(v3(0, 0, 1));
  const float theta_x = fminf(acosf(dot3(dir, x_dir)), acosf(dot3(dir, -x_dir)));
  const float theta_y = fminf(acosf(dot3(dir, y_dir)), acosf(dot3(dir, -y_dir)));
  const float theta_z = fminf(acosf(dot3(dir, z_dir)), acosf(dot3(dir, -z_dir)));
  float step = len;
  const bool het = p.heterogeneous != 0;
  DdaWalk walk;
  float d_prev = 0.f, e_prev = 0.f, d_cur = 0.f, e_cur = 0.f;
  float mean = p.extinction_c;
  int current_step = 1;
  if (het) {
    walk.begin(p.grid, xyz(start), xyz(end));
    walk.next(p.grid, d_prev, e_prev);
    mean = e_prev;
  }
  const bool ada = (b.flags & 1u) != 0;
  // BeamRadiance.h:449-455 always looks the energy up in beams_const[id]
  float en[3] = {0.f, 0.f, 0.f};
  if (WRITE && i < p.n_const) { en[0] = p.beams[i].energy[0]; en[1] = p.beams[i].energy[1]; en[2] = p.beams[i].energy[2]; }
  V3 prior = v3(start.x, start.y, start.z);
  uint32_t count = 0;
  const uint32_t base = WRITE ? p.seg_offset[g] : 0u;
  while (j < len && len > 1e-6) {
    if (het) {
      if (!walk.next(p.grid, d_cur, e_cur)) break;   // cannot happen: the last sample ends the loop below
      step = d_cur - d_prev;
      mean = mean + ((e_cur - mean) / (current_step + 1));
    }
    const V3 next = prior + (dir * step);
    const V3 s_start = prior;
    V3 s_end;
    if (len <= length3(next - xyz(start))) { j = fabsf(len); s_end = xyz(end); }
    else { j = length3(next - xyz(start)); s_end = next; }
    prior = s_end;
    float radius;
    if (ada) {
      const V3 ou = v3(b.omega_u[0], b.omega_u[1], b.omega_u[2]);
      const V3 omega_u = normalize3(ou);
      const float cos_theta = dot3(dir, omega_u);
      const float hyp_length = j / cos_theta;
      const V4 end_u = start + (v4(b.omega_u[0], b.omega_u[1], b.omega_u[2], b.omega_u[3]) * hyp_length);
      radius = length3(s_end - xyz(end_u));
    } else radius = b.radius;
    if (WRITE) {
      float mnx = fminf(s_start.x, s_end.x), mny = fminf(s_start.y, s_end.y), mnz = fmaxf(s_start.z, s_end.z);
      float mxx = fmaxf(s_start.x, s_end.x), mxy = fmaxf(s_start.y, s_end.y), mxz = fminf(s_start.z, s_end.z);
      if (len > 0.001) {
        float th; int axis;
        if (theta_x < theta_y && theta_x < theta_z) { th = theta_x; axis = 0; }
        else if (theta_y < theta_z) { th = theta_y; axis = 1; }
        else { th = theta_z; axis = 2; }
        const float hypotenuse = radius / sinf(th);
        const float adj = radius / cosf(th);
        const float height = adj * sinf(th);
        const float px = (axis == 0) ? height : hypotenuse;
        const float py = (axis == 1) ? height : hypotenuse;
        const float pz = (axis == 2) ? height : hypotenuse;
        mnx = mnx - px; mny = mny - py; mnz = mnz + pz;
        mxx = mxx + px; mxy = mxy + py; mxz = mxz - pz;
      }
      pb_segment s;
      s.start[0] = s_start.x; s.start[1] = s_start.y; s.start[2] = s_start.z;
      s.end[0] = s_end.x; s.end[1] = s_end.y; s.end[2] = s_end.z;
      s.orig_start[0] = start.x; s.orig_start[1] = start.y; s.orig_start[2] = start.z;
      s.radius = radius;
      // PhotonSeg.min/max store z flipped; the gate uses min_b = (min.x, min.y, max.z)
      s.cull_lo[0] = mnx; s.cull_lo[1] = mny; s.cull_lo[2] = mxz;
      s.cull_hi[0] = mxx; s.cull_hi[1] = mxy; s.cull_hi[2] = mnz;
      s.energy[0] = en[0]; s.energy[1] = en[1]; s.energy[2] = en[2];
      s.s_ext = het ? e_prev : 0.f; s.e_ext = het ? e_cur : 0.f; s.c_ext = mean;
      s.seg_id = current_step; s.id = i;
      s.flags = ada ? 1u : 0u;
      s.reserved[0] = (g >= p.n_const) ? 1u : 0u;   // which list the beam is in; PhotonSeg.scattered itself stays false
      s.reserved[1] = s.reserved[2] = 0u;
      p.segs[base + count] = s;
    }
    count++;
    if (het) {
      if (walk.state == 2) j = len + 1;   // that was the last sample (CastPhotons.h:273-275)
      current_step++;
      d_prev = d_cur; e_prev = e_cur;
    }
  }
  if (!WRITE) p.seg_count[g] = count;
}

// Root box of CastPhotonBeams (CastPhotons.h:507-512, :384-387): bounds of all beam end points.  Blocks write
// one partial box each (grid-stride); a second launch of one block over those partials gives the result.
constexpr int ROOT_BOX_BLOCKS = 256;
__global__ __launch_bounds__(1024) void root_box_kernel(const float* __restrict__ pts6, int stride_floats, int n,
                                                        float* __restrict__ out6) {
  __shared__ float red[6][1024 / 64];
  float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  // record i holds two points: floats 0..2 and 4..6 of a beam (start, end), or lo / hi of a partial box
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* r = pts6 + (size_t)i * stride_floats;
    const int second = (stride_floats == 6) ? 3 : 4;
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], fminf(r[k], r[second + k]));
      hi[k] = fmaxf(hi[k], fmaxf(r[k], r[second + k]));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_down(lo[k], off, 64));
      hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off, 64));
    }
    if (lane == 0) { red[k][wave] = lo[k]; red[3 + k][wave] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = red[k][0];
    for (int w = 1; w < 1024 / 64; ++w) v = (k < 3) ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
    out6[(size_t)blockIdx.x * 6 + k] = v;
  }
}

// Scratch of the photon pass lives in the context and only grows: a stop-motion loop re-casts every frame,
// and an allocation or a free costs more than the kernels of a small pass.
int emit_scratch(pb_ctx* ctx, int which, size_t words, uint32_t** out) {
  if (words > ctx->emit_tmp_words[which]) {
    if (ctx->d_emit_tmp[which]) PB_HIP(hipFree(ctx->d_emit_tmp[which]));
    ctx->d_emit_tmp[which] = nullptr; ctx->emit_tmp_words[which] = 0;
    const size_t want = words + words / 4;
    PB_HIP(hipMalloc(&ctx->d_emit_tmp[which], sizeof(uint32_t) * want));
    ctx->emit_tmp_words[which] = want;
  }
  *out = ctx->d_emit_tmp[which];
  return PB_OK;
}

}  // namespace

int pb_emit_run(pb_ctx* ctx, int n_photons, uint64_t seed, bool recast) {
  // recast: n_photons is the number of beams of the first list, which are kept and cast again
  pb_beam* old_beams = recast ? ctx->d_beams : nullptr;
  if (ctx->ctl.heterogeneous && !ctx->d_grid) { pb_set_error("pb_emit_beams: HETEROGENEOUS needs pb_set_grid first"); return PB_ERR_STATE; }
  hipStream_t s = ctx->stream;
  {
    // the pass rewrites the segment records in place; frames of the old set may still be running on the caller's streams
    const int rc = pb_wait_for_readers(ctx, s, true);
    if (rc != PB_OK) return rc;
  }
  PB_HIP(hipEventRecord(ctx->ev[6], s));
  // scratch: cnt_const[n] cnt_scatt[n] off_const[n] off_scatt[n] totals[4] dropped[1] root[6] + scan tmp
  const size_t n = (size_t)n_photons;
  const size_t scan_words = pb_scan_tmp_words(n_photons > 0 ? 2 * n_photons : 1);
  const size_t head = 4 * n + 16 + 6 * ROOT_BOX_BLOCKS;
  uint32_t* tmp_a = nullptr;
  int rc = emit_scratch(ctx, 0, head + scan_words, &tmp_a);
  if (rc != PB_OK) return rc;
  PB_HIP(hipMemsetAsync(tmp_a, 0, sizeof(uint32_t) * (head + scan_words), s));
  uint32_t* cnt_const = tmp_a; uint32_t* cnt_scatt = tmp_a + n;
  uint32_t* off_const = tmp_a + 2 * n; uint32_t* off_scatt = tmp_a + 3 * n;
  uint32_t* totals = tmp_a + 4 * n;            // [0] n_const [1] n_scatt [2] n_seg [3] dropped
  float* d_root = (float*)(tmp_a + 4 * n + 8); // 6 floats
  float* d_root_partial = (float*)(tmp_a + 4 * n + 16);   // 6 floats per block of the first root-box launch
  uint32_t* scan_tmp = tmp_a + head;

  EmitParams ep;
  std::memset(&ep, 0, sizeof(ep));
  ep.tris = ctx->d_tris; ep.n_tri = ctx->n_tri;
  for (int k = 0; k < 16; ++k) ep.matrix[k] = ctx->matrix[k];
  ep.ctl = ctx->ctl; ep.seed = seed; ep.n_photons = n_photons;
  ep.cnt_const = cnt_const; ep.cnt_scatt = cnt_scatt; ep.off_const = off_const; ep.off_scatt = off_scatt;
  ep.dropped = totals + 3;
  ep.grid.z = ctx->d_grid; ep.grid.side = ctx->grid_side; ep.grid.fast_div = ctx->grid_fast_div;
  ep.sources = old_beams;
  const int blocks = (n_photons + EMIT_THREADS - 1) / EMIT_THREADS;
  // Up to EMIT_STAGE_MAX_PHOTONS photons the beams are cast once, into slots of the photon's own, and moved when the
  // offsets are known (the photon pass of a small set is one long chain of dependent steps per lane; casting twice,
  // to count and then to write, doubled it).  A photon that outgrows its slots sends the pass the two-pass way.
  const bool staged = n_photons > 0 && n_photons <= EMIT_STAGE_MAX_PHOTONS && !pb_tune().emit_twice;
  pb_beam* stage = nullptr;
  const int stage_slots = emit_stage_slots(ctx->ctl.bounces);
  if (staged) {
    uint32_t* raw = nullptr;
    rc = emit_scratch(ctx, 2, (sizeof(pb_beam) * 2 * stage_slots * n + sizeof(uint32_t) - 1) / sizeof(uint32_t), &raw);
    if (rc != PB_OK) return rc;
    stage = (pb_beam*)raw;
    ep.staged = stage_slots; ep.overflow = totals + 4;
    ep.beams_const = stage; ep.beams_scatt = stage;
    if (recast) hipLaunchKernelGGL((emit_kernel<true, true>), dim3(blocks), dim3(EMIT_THREADS), 0, s, ep);
    else hipLaunchKernelGGL((emit_kernel<true, false>), dim3(blocks), dim3(EMIT_THREADS), 0, s, ep);
  } else if (n_photons > 0) {
    if (recast) hipLaunchKernelGGL((emit_kernel<false, true>), dim3(blocks), dim3(EMIT_THREADS), 0, s, ep);
    else hipLaunchKernelGGL((emit_kernel<false, false>), dim3(blocks), dim3(EMIT_THREADS), 0, s, ep);
  }
  PB_HIP(hipGetLastError());
  // one scan over both count arrays (they are adjacent, and so are the offset arrays): a photon's offset into the
  // scattered list then already counts the first list before it, whose length is the first scattered offset
  rc = pb_exclusive_scan_u32(cnt_const, off_const, 2 * n_photons, totals + 1, scan_tmp, s);
  if (rc != PB_OK) return rc;
  uint32_t h_tot[4] = {0, 0, 0, 0};
  PB_HIP(hipMemcpyAsync(h_tot + 1, totals + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  if (n_photons > 0) PB_HIP(hipMemcpyAsync(h_tot + 0, off_scatt, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  uint32_t h_overflow = 0;
  if (staged) PB_HIP(hipMemcpyAsync(&h_overflow, totals + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  const int listed_const = (int)h_tot[0];                  // beams the count pass put in the first list
  const int nc = recast ? n_photons : listed_const, ns = (int)h_tot[1] - listed_const, nb = nc + ns;

  // the new list goes into the spare buffer (a re-cast reads the old list while it writes the new one)
  const size_t want_beams = (size_t)(nb > 0 ? nb : 1);
  if (want_beams > ctx->cap_beams_spare) {
    if (ctx->d_beams_spare) PB_HIP(hipFree(ctx->d_beams_spare));
    ctx->d_beams_spare = nullptr; ctx->cap_beams_spare = 0;
    PB_HIP(hipMalloc(&ctx->d_beams_spare, sizeof(pb_beam) * (want_beams + want_beams / 8)));
    ctx->cap_beams_spare = want_beams + want_beams / 8;
  }
  pb_beam* new_beams = ctx->d_beams_spare;
  if (recast && nc > 0)
    PB_HIP(hipMemcpyAsync(new_beams, old_beams, sizeof(pb_beam) * (size_t)nc, hipMemcpyDeviceToDevice, s));
  ep.beams_const = new_beams; ep.beams_scatt = new_beams + nc - listed_const;   // off_scatt counts the first list's beams
  if (staged && !h_overflow) {
    hipLaunchKernelGGL(gather_staged_kernel, dim3((n_photons + 255) / 256), dim3(256), 0, s, stage, cnt_const, cnt_scatt, off_const,
                       off_scatt, ep.beams_const, ep.beams_scatt, n_photons, stage_slots);
  } else if (n_photons > 0) {
    PB_HIP(hipMemsetAsync(totals + 3, 0, sizeof(uint32_t), s));   // the pass before this one reported the same drops
    ep.staged = 0;
    if (recast) hipLaunchKernelGGL((emit_kernel<true, true>), dim3(blocks), dim3(EMIT_THREADS), 0, s, ep);
    else hipLaunchKernelGGL((emit_kernel<true, false>), dim3(blocks), dim3(EMIT_THREADS), 0, s, ep);
  }
  PB_HIP(hipGetLastError());
  // adopt the new list; the old one becomes the spare (stream order keeps its readers ahead of the next writer)
  std::swap(ctx->d_beams, ctx->d_beams_spare);
  std::swap(ctx->cap_beams, ctx->cap_beams_spare);
  ctx->n_beams_const = nc; ctx->n_beams_scatt = ns;

  // segments
  uint32_t* seg_buf = nullptr;
  const size_t scan_words2 = pb_scan_tmp_words(nb > 0 ? nb : 1);
  rc = emit_scratch(ctx, 1, 2 * (size_t)(nb > 0 ? nb : 1) + scan_words2, &seg_buf);
  if (rc != PB_OK) return rc;
  uint32_t* seg_count = seg_buf; uint32_t* seg_offset = seg_buf + (nb > 0 ? nb : 1);
  uint32_t* scan_tmp2 = seg_buf + 2 * (size_t)(nb > 0 ? nb : 1);
  BoundParams bp;
  std::memset(&bp, 0, sizeof(bp));
  bp.beams = ctx->d_beams; bp.n_const = nc; bp.n_scatt = ns;
  bp.extinction_c = ctx->ctl.absorption_c + ctx->ctl.scattering_c;
  bp.heterogeneous = ctx->ctl.heterogeneous ? 1 : 0;
  bp.grid.z = ctx->d_grid; bp.grid.side = ctx->grid_side; bp.grid.fast_div = ctx->grid_fast_div;
  bp.seg_count = seg_count; bp.seg_offset = seg_offset;
  if (nb > 0) hipLaunchKernelGGL(bound_kernel<false>, dim3((nb + 255) / 256), dim3(256), 0, s, bp);
  PB_HIP(hipGetLastError());
  rc = pb_exclusive_scan_u32(seg_count, seg_offset, nb, totals + 2, scan_tmp2, s);
  if (rc != PB_OK) return rc;
  if (nb > 0) {
    static_assert(sizeof(pb_beam) % sizeof(float) == 0 && offsetof(pb_beam, start) == 0 && offsetof(pb_beam, end) == 16,
                  "root_box_kernel reads start and end as floats 0..2 and 4..6 of a beam record");
    int rb = (nb + 1023) / 1024;
    if (rb > ROOT_BOX_BLOCKS) rb = ROOT_BOX_BLOCKS;
    hipLaunchKernelGGL(root_box_kernel, dim3(rb), dim3(1024), 0, s, (const float*)ctx->d_beams,
                       (int)(sizeof(pb_beam) / sizeof(float)), nb, d_root_partial);
    hipLaunchKernelGGL(root_box_kernel, dim3(1), dim3(1024), 0, s, d_root_partial, 6, rb, d_root);
  }
  PB_HIP(hipGetLastError());
  float h_root[6] = {0, 0, 0, 0, 0, 0};
  PB_HIP(hipMemcpyAsync(h_tot, totals, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(h_root, d_root, sizeof(float) * 6, hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  const int nseg = (int)h_tot[2];
  rc = pb_ensure_segment_capacity(ctx, nseg);
  if (rc != PB_OK) return rc;
  bp.segs = ctx->d_segs;
  if (nb > 0) hipLaunchKernelGGL(bound_kernel<true>, dim3((nb + 255) / 256), dim3(256), 0, s, bp);
  PB_HIP(hipGetLastError());
  ctx->n_seg = nseg;
  for (int k = 0; k < 3; ++k) { ctx->root_lo[k] = h_root[k]; ctx->root_hi[k] = h_root[3 + k]; }
  rc = pb_prepare_frames(ctx, pb_default_mode(ctx), s);
  if (rc != PB_OK) return rc;
  PB_HIP(hipEventRecord(ctx->ev[7], s));
  ctx->ev_valid[3] = true;
  PB_HIP(hipStreamSynchronize(s));
  ctx->have_segments = true;
  ctx->segs_version++;
  if (h_tot[3] != 0) {
    pb_set_error("pb_emit_beams: " + std::to_string(h_tot[3]) + " nested scattering events exceeded the per-photon stack and were dropped");
    return PB_ERR_CAPACITY;
  }
  return PB_OK;
}
