// See photon_beams_host.h.  Host glue only: every computation of the path happens in libphoton_beams.so.
#include "photon_beams_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <chrono>
#include <deque>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>

namespace pbhost {

vec4 light_position{0, (float)-0.8, (float)-0.5, 1};
vec3 light_power{0.5f, 0.5f, 0.5f};
float absorption_c = (float)0.035;
float scattering_c = (float)0.005;
float extinction_c = absorption_c + scattering_c;
vec4 camera{0, 0, -3, 1.0f};
vec3 theta{0.0f, 0.0f, 0.0f};
Grid GRID;
pb_control control = [] { pb_control c; pb_control_defaults(&c); return c; }();
std::vector<int> devices{0};

namespace {

pb_ctx* g_ctx = nullptr;
std::vector<pb_beam> g_beams;      // beams_const list, then beams_scattered list
int g_n_const = 0, g_n_scatt = 0;
std::vector<pb_segment> g_segs;    // segments of the last photon pass, const part then scattered part
int g_bound_calls = 0;
std::vector<Triangle> g_triangles;

void check(int rc, const char* what) {
  if (rc != PB_OK) throw std::runtime_error(std::string(what) + ": " + pb_last_error());
}

pb_control current_control() {
  pb_control c = control;
  std::memcpy(c.light_position, &light_position, sizeof(float) * 4);
  std::memcpy(c.light_power, &light_power, sizeof(float) * 3);
  c.absorption_c = absorption_c; c.scattering_c = scattering_c;
  std::memcpy(c.camera, &camera, sizeof(float) * 4);
  std::memcpy(c.theta, &theta, sizeof(float) * 3);
  return c;
}

// Contexts of devices[1..]: they only render.  What they need - scene, grid, rendered segment set - is kept on the
// host and uploaded when it changes (g_*_version against the versions a helper has seen).
struct Helper { pb_ctx* ctx = nullptr; int device = 0; unsigned scene_seen = 0, segs_seen = 0; };
std::vector<Helper> g_helpers;
std::vector<float> g_tri_flat;          // 9 floats per triangle, as pb_set_scene takes them
float g_matrix[16];
std::vector<pb_segment> g_render_segs;  // the rendered set (BuildTree / LoadSegments) and its root box
float g_render_lo[3], g_render_hi[3];
unsigned g_scene_version = 0, g_segs_version = 0;

void drop_helpers() {
  for (Helper& h : g_helpers) if (h.ctx) pb_destroy(h.ctx);
  g_helpers.clear();
}

pb_ctx* context() {
  if (devices.empty()) devices.push_back(0);
  if (!g_ctx) check(pb_create(&control, devices[0], &g_ctx), "pb_create");
  return g_ctx;
}

void upload_grid(pb_ctx* ctx, const Grid& grid);
void remember_scene(const std::vector<float>& tri, const mat4& matrix);

// The helper contexts, one per entry of devices[1..], brought up to date with scene, grid and segment set.
std::vector<Helper>& helpers() {
  bool same = g_helpers.size() + 1 == devices.size();
  for (size_t i = 0; same && i < g_helpers.size(); ++i) same = g_helpers[i].device == devices[i + 1];
  if (!same) {
    drop_helpers();
    for (size_t i = 1; i < devices.size(); ++i) {
      Helper h;
      h.device = devices[i];
      check(pb_create(&control, h.device, &h.ctx), "pb_create (helper device)");
      g_helpers.push_back(h);
    }
  }
  for (Helper& h : g_helpers) {
    if (h.scene_seen != g_scene_version) {
      check(pb_set_scene(h.ctx, g_tri_flat.data(), (int)(g_tri_flat.size() / 9), g_matrix), "pb_set_scene (helper device)");
      if (control.heterogeneous) upload_grid(h.ctx, GRID);
      h.scene_seen = g_scene_version;
    }
    if (h.segs_seen != g_segs_version) {
      if (control.heterogeneous) upload_grid(h.ctx, GRID);      // stop motion: a new field comes with a new set
      check(pb_upload_segments(h.ctx, g_render_segs.data(), (int)g_render_segs.size(), g_render_lo, g_render_hi),
            "pb_upload_segments (helper device)");
      h.segs_seen = g_segs_version;
    }
  }
  return g_helpers;
}

void remember_scene(const std::vector<float>& tri, const mat4& matrix) {
  g_tri_flat = tri;
  std::memcpy(g_matrix, &matrix.m[0][0], sizeof(g_matrix));
  g_scene_version++;
}

void remember_render_set(const pb_segment* segs, size_t n, const float lo[3], const float hi[3]) {
  g_render_segs.assign(segs, segs + n);
  for (int k = 0; k < 3; ++k) { g_render_lo[k] = lo[k]; g_render_hi[k] = hi[k]; }
  g_segs_version++;
}

// One frame into scr->buffer: on one device a single call; on several, bands of 64 rows dealt round robin (the lit
// cone makes rows cost very differently), one host thread per device, each writing its rows of the host image.
void render_frame(screen* scr, int mode) {
  pb_ctx* ctx = context();
  {
    // skeleton.cpp:333-334 (DrawStandard), :370-371 (DrawHet), :417-418 (Draw): the matrix of the frame is made from
    // the globals when the frame starts; the beams keep the space they were cast in
    mat4 now;
    TransformationMatrix(now);
    if (!g_tri_flat.empty() && std::memcmp(g_matrix, &now.m[0][0], sizeof(g_matrix)) != 0) {
      check(pb_set_scene(ctx, g_tri_flat.data(), (int)(g_tri_flat.size() / 9), &now.m[0][0]), "pb_set_scene");
      remember_scene(g_tri_flat, now);
    }
  }
  const int w = scr->width, h = scr->height;
  std::vector<float> rgb((size_t)w * h * 3);
  if (devices.size() <= 1) {
    check(pb_render(ctx, 0, 0, w, h, mode, /*screen bins*/ 2, rgb.data()), "pb_render");
  } else {
    std::vector<Helper>& hs = helpers();
    std::vector<pb_ctx*> all{ctx};
    for (Helper& x : hs) all.push_back(x.ctx);
    const int n_dev = (int)all.size(), band = 64, n_bands = (h + band - 1) / band;
    std::vector<std::string> errors((size_t)n_dev);
    std::vector<std::thread> workers;
    for (int d = 0; d < n_dev; ++d)
      workers.emplace_back([&, d] {
        for (int b = d; b < n_bands; b += n_dev) {
          const int y0 = b * band, hb = (y0 + band <= h) ? band : h - y0;
          if (pb_render(all[(size_t)d], 0, y0, w, hb, mode, 2, rgb.data() + (size_t)y0 * w * 3) != PB_OK) {
            errors[(size_t)d] = pb_last_error();     // the error text is per thread: take it here
            return;
          }
        }
      });
    for (std::thread& t : workers) t.join();
    for (const std::string& e : errors) if (!e.empty()) throw std::runtime_error("pb_render: " + e);
  }
  check(pb_quantise_argb(rgb.data(), w * h, scr->buffer), "pb_quantise_argb");
}

void recreate_context_if_control_changed() {
  static pb_control last = control;
  pb_control now = current_control();
  if (g_ctx && std::memcmp(&last, &now, sizeof(now)) != 0) { pb_destroy(g_ctx); g_ctx = nullptr; drop_helpers(); }
  control = now;
  last = now;
}

vec4 v4(const float* p) { return vec4{p[0], p[1], p[2], p[3]}; }

void upload_grid(pb_ctx* ctx, const Grid& grid) {
  if (grid.side_points < 2 || (int)grid.geometric_points.size() != grid.side_points * grid.side_points)
    throw std::runtime_error("GRID is empty: call CreateSurface before a HETEROGENEOUS photon pass");
  std::vector<float> sigma(grid.geometric_points.size());
  for (size_t i = 0; i < sigma.size(); ++i) sigma[i] = grid.geometric_points[i].z;
  check(pb_set_grid(ctx, sigma.data(), grid.side_points), "pb_set_grid");
}

}  // namespace

void LoadTestModel(std::vector<Triangle>& triangles) {
  float tri[30 * 9], nrm[30 * 3];
  int n = pb_load_test_model(tri, nrm, 30);
  if (n != 30) throw std::runtime_error("pb_load_test_model failed");
  // TestModelH.h:51-57, :101-184: white ceiling/floor/back, green left, red right, red short block, blue tall block
  const vec3 red{0.75f, 0.15f, 0.15f}, green{0.15f, 0.75f, 0.15f}, blue{0.15f, 0.15f, 0.75f}, white{0.75f, 0.75f, 0.75f};
  triangles.clear();
  for (int i = 0; i < n; ++i) {
    Triangle t;
    t.v0 = vec4{tri[i * 9 + 0], tri[i * 9 + 1], tri[i * 9 + 2], 1.0f};
    t.v1 = vec4{tri[i * 9 + 3], tri[i * 9 + 4], tri[i * 9 + 5], 1.0f};
    t.v2 = vec4{tri[i * 9 + 6], tri[i * 9 + 7], tri[i * 9 + 8], 1.0f};
    t.normal = vec4{nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2], 1.0f};
    t.colour = (i < 4) ? white : (i < 6) ? green : (i < 8) ? red : (i < 10) ? white : (i < 20) ? red : blue;
    t.water = false; t.reflective = false;
    triangles.push_back(t);
  }
}

void TransformationMatrix(mat4& M) {
  float th[3] = {theta.x, theta.y, theta.z}, cam[4] = {camera.x, camera.y, camera.z, camera.w};
  pb_transformation_matrix(th, cam, &M.m[0][0]);
}

AABB CastPhotonBeams(int number, std::vector<PhotonBeam>& beams_const, std::vector<PhotonBeam>& beams_scattered,
                     const mat4& matrix, const std::vector<Triangle>& triangles) {
  extinction_c = absorption_c + scattering_c;
  recreate_context_if_control_changed();
  pb_ctx* ctx = context();
  std::vector<float> tri(triangles.size() * 9);
  for (size_t i = 0; i < triangles.size(); ++i) {
    const vec4* v[3] = {&triangles[i].v0, &triangles[i].v1, &triangles[i].v2};
    for (int k = 0; k < 3; ++k) { tri[i * 9 + k * 3] = v[k]->x; tri[i * 9 + k * 3 + 1] = v[k]->y; tri[i * 9 + k * 3 + 2] = v[k]->z; }
  }
  g_triangles = triangles;
  check(pb_set_scene(ctx, tri.data(), (int)triangles.size(), &matrix.m[0][0]), "pb_set_scene");
  remember_scene(tri, matrix);
  if (control.heterogeneous) upload_grid(ctx, GRID);
  int n_beams = 0, n_segs = 0;
  check(pb_emit_beams(ctx, number, control.seed, &n_beams, &n_segs), "pb_emit_beams");
  check(pb_beam_counts(ctx, &g_n_const, &g_n_scatt), "pb_beam_counts");
  g_beams.resize((size_t)(n_beams > 0 ? n_beams : 1));
  int got = 0;
  check(pb_download_beams(ctx, g_beams.data(), (int)g_beams.size(), &got), "pb_download_beams");
  g_beams.resize((size_t)got);
  g_segs.resize((size_t)(n_segs > 0 ? n_segs : 1));
  check(pb_download_segments(ctx, g_segs.data(), (int)g_segs.size(), &got), "pb_download_segments");
  g_segs.resize((size_t)got);
  g_bound_calls = 0;
  for (int i = 0; i < (int)g_beams.size(); ++i) {
    const pb_beam& b = g_beams[i];
    PhotonBeam o;
    o.start = v4(b.start); o.end = v4(b.end); o.omega_u = v4(b.omega_u); o.omega_v = v4(b.omega_u);
    o.ada_width = (b.flags & 1u) != 0; o.offset = 0.0f; o.radius = b.radius;
    o.energy = dvec3{b.energy[0], b.energy[1], b.energy[2]};
    o.index_hit = b.index_hit; o.absorbed = (b.flags & 4u) != 0; o.scattered = (b.flags & 2u) != 0;
    (i < g_n_const ? beams_const : beams_scattered).push_back(o);
  }
  float lo[3], hi[3];
  check(pb_get_root_box(ctx, lo, hi), "pb_get_root_box");
  AABB root;   // CastPhotons.h:384-387: z stored flipped
  root.min = vec4{lo[0], lo[1], hi[2], 1}; root.max = vec4{hi[0], hi[1], lo[2], 1};
  root.mid = vec4{(root.min.x + root.max.x) / 2.0f, (root.min.y + root.max.y) / 2.0f, (root.min.z + root.max.z) / 2.0f, 1};
  return root;
}

namespace {
void beams_to_host(const std::vector<pb_beam>& src, size_t from, size_t to, std::vector<PhotonBeam>& out) {
  for (size_t i = from; i < to; ++i) {
    const pb_beam& b = src[i];
    PhotonBeam o;
    o.start = v4(b.start); o.end = v4(b.end); o.omega_u = v4(b.omega_u); o.omega_v = v4(b.omega_u);
    o.ada_width = (b.flags & 1u) != 0; o.offset = 0.0f; o.radius = b.radius;
    o.energy = dvec3{b.energy[0], b.energy[1], b.energy[2]};
    o.index_hit = b.index_hit; o.absorbed = (b.flags & 4u) != 0; o.scattered = (b.flags & 2u) != 0;
    out.push_back(o);
  }
}
unsigned long long g_recasts = 0;
}  // namespace

AABB RecastPhotonBeams(std::vector<PhotonBeam>&, std::vector<PhotonBeam>& beams_scattered, const mat4&,
                       const std::vector<Triangle>&) {
  pb_ctx* ctx = context();
  if (control.heterogeneous) upload_grid(ctx, GRID);
  int n_scatt = 0, n_segs = 0;
  // a new random stream per call, as the reference's generator simply runs on
  check(pb_recast_beams(ctx, control.seed + 0x9E3779B97F4A7C15ull * ++g_recasts, &n_scatt, &n_segs), "pb_recast_beams");
  check(pb_beam_counts(ctx, &g_n_const, &g_n_scatt), "pb_beam_counts");
  g_beams.resize((size_t)(g_n_const + g_n_scatt > 0 ? g_n_const + g_n_scatt : 1));
  int got = 0;
  check(pb_download_beams(ctx, g_beams.data(), (int)g_beams.size(), &got), "pb_download_beams");
  g_beams.resize((size_t)got);
  g_segs.resize((size_t)(n_segs > 0 ? n_segs : 1));
  check(pb_download_segments(ctx, g_segs.data(), (int)g_segs.size(), &got), "pb_download_segments");
  g_segs.resize((size_t)got);
  g_bound_calls = 0;
  beams_scattered.clear();
  beams_to_host(g_beams, (size_t)g_n_const, g_beams.size(), beams_scattered);
  float lo[3], hi[3];
  check(pb_get_root_box(ctx, lo, hi), "pb_get_root_box");
  AABB root;
  root.min = vec4{lo[0], lo[1], hi[2], 1}; root.max = vec4{hi[0], hi[1], lo[2], 1};
  root.mid = vec4{(root.min.x + root.max.x) / 2.0f, (root.min.y + root.max.y) / 2.0f, (root.min.z + root.max.z) / 2.0f, 1};
  return root;
}

void BoundPhotonBeams(std::vector<PhotonBeam>& beams, std::vector<PhotonSeg>& items, const std::vector<Triangle>&) {
  // segments of the const list come first (CastPhotons.h:227 numbers each list from 0, so ids cannot tell the
  // lists apart); the library marks the scattered list's segments in reserved[0]
  size_t split = g_segs.size();
  for (size_t i = 0; i < g_segs.size(); ++i)
    if (g_segs[i].reserved[0] != 0u) { split = i; break; }
  const bool first = (g_bound_calls++ == 0) && ((int)beams.size() == g_n_const);
  const size_t a = first ? 0 : split, b = first ? split : g_segs.size();
  for (size_t i = a; i < b; ++i) {
    const pb_segment& s = g_segs[i];
    PhotonSeg o;
    o.start = vec4{s.start[0], s.start[1], s.start[2], 1}; o.end = vec4{s.end[0], s.end[1], s.end[2], 1};
    o.orig_start = vec4{s.orig_start[0], s.orig_start[1], s.orig_start[2], 1};
    o.min = vec4{s.cull_lo[0], s.cull_lo[1], s.cull_hi[2], 1};   // PhotonSeg.min/max keep z flipped
    o.max = vec4{s.cull_hi[0], s.cull_hi[1], s.cull_lo[2], 1};
    o.mid = vec4{(o.min.x + o.max.x) / 2.0f, (o.min.y + o.max.y) / 2.0f, (o.min.z + o.max.z) / 2.0f, 1};
    o.radius = s.radius; o.ada_width = (s.flags & 1u) != 0; o.id = s.id;
    o.s_ext = s.s_ext; o.e_ext = s.e_ext; o.c_ext = s.c_ext; o.seg_id = s.seg_id; o.scattered = (s.flags & 2u) != 0;
    items.push_back(o);
  }
}

Node* newNode(AABB data) {
  Node* n = new Node;
  n->aabb = data; n->leaf = false;
  return n;
}

void BuildTree(Node* parent, std::vector<PhotonSeg>& child, int) {
  pb_ctx* ctx = context();
  std::vector<pb_segment> segs(child.size());
  for (size_t i = 0; i < child.size(); ++i) {
    const PhotonSeg& s = child[i];
    pb_segment& d = segs[i];
    std::memset(&d, 0, sizeof(d));
    d.start[0] = s.start.x; d.start[1] = s.start.y; d.start[2] = s.start.z;
    d.end[0] = s.end.x; d.end[1] = s.end.y; d.end[2] = s.end.z;
    d.orig_start[0] = s.orig_start.x; d.orig_start[1] = s.orig_start.y; d.orig_start[2] = s.orig_start.z;
    d.radius = s.radius;
    d.cull_lo[0] = s.min.x; d.cull_lo[1] = s.min.y; d.cull_lo[2] = s.max.z;
    d.cull_hi[0] = s.max.x; d.cull_hi[1] = s.max.y; d.cull_hi[2] = s.min.z;
    // BeamRadiance.h:449-455 with seg.scattered always false (CastPhotons.h:398): beams_const[seg.id]
    const int list0 = s.scattered ? g_n_const : 0, listn = s.scattered ? g_n_scatt : g_n_const;
    if (s.id >= 0 && s.id < listn) std::memcpy(d.energy, g_beams[(size_t)(list0 + s.id)].energy, sizeof(float) * 3);
    d.s_ext = s.s_ext; d.e_ext = s.e_ext; d.c_ext = s.c_ext; d.seg_id = s.seg_id; d.id = s.id;
    d.flags = (s.ada_width ? 1u : 0u) | (s.scattered ? 2u : 0u);
  }
  const AABB& r = parent->aabb;   // un-flip: min_b = (min.x, min.y, max.z), Intersections.h:85-86
  const float lo[3] = {r.min.x, r.min.y, r.max.z}, hi[3] = {r.max.x, r.max.y, r.min.z};
  check(pb_upload_segments(ctx, segs.data(), (int)segs.size(), lo, hi), "pb_upload_segments");
  check(pb_build_lbvh(ctx), "pb_build_lbvh");
  remember_render_set(segs.data(), segs.size(), lo, hi);
  parent->leaf = child.size() <= 2;
}

void CreateSurface(int triangle_number, float height, double time) {
  const int side = triangle_number + 1;
  std::vector<float> pts((size_t)side * side * 4);
  const int n = pb_create_surface(triangle_number, height, time, pts.data(), side * side);
  if (n != side * side) throw std::runtime_error("pb_create_surface failed");
  GRID.height = height;
  GRID.side_points = side;
  GRID.geometric_points.resize((size_t)n);
  for (int i = 0; i < n; ++i) GRID.geometric_points[(size_t)i] = v4(&pts[(size_t)i * 4]);
}

float Extinction3D(const vec4 start, const vec4 end, const Grid grid, std::vector<vec2>& dist_ext) {
  pb_ctx* ctx = context();
  upload_grid(ctx, grid);
  const float a[3] = {start.x, start.y, start.z}, b[3] = {end.x, end.y, end.z};
  const int cap = 4 * grid.side_points + 16;   // at most one sample per cell face crossed, plus the two ends
  std::vector<float> samples((size_t)cap * 2);
  float mean = 0.0f;
  int count = 0;
  check(pb_extinction3d(ctx, a, b, 1, &mean, &count, samples.data(), cap), "pb_extinction3d");
  for (int i = 0; i < count && i < cap; ++i) dist_ext.push_back(vec2{samples[(size_t)i * 2], samples[(size_t)i * 2 + 1]});
  return mean;
}

screen* InitializeSDL(int width, int height, bool) {
  screen* s = new screen;
  s->width = width; s->height = height;
  s->buffer = new uint32_t[(size_t)width * height];
  std::memset(s->buffer, 0, sizeof(uint32_t) * (size_t)width * height);
  return s;
}

void PutPixelSDL(screen* s, int x, int y, vec3 colour) {
  if (x < 0 || x >= s->width || y < 0 || y >= s->height) return;
  const float rgb[3] = {colour.x, colour.y, colour.z};
  pb_quantise_argb(rgb, 1, &s->buffer[(size_t)y * s->width + x]);
}

void SDL_Renderframe(screen*) {}

// 32-bit BMP with BI_BITFIELDS masks (A 0xff000000, R 0x00ff0000, G 0x0000ff00, B 0x000000ff), bottom-up,
// as SDL_SaveBMP writes an ARGB8888 surface (pixels-per-metre fields zero): the header equals the reference's
// screenshot.bmp byte for byte.
void SDL_SaveImage(screen* s, const char* filename) {
  const uint32_t w = (uint32_t)s->width, h = (uint32_t)s->height, data = w * h * 4, off = 14 + 108;
  unsigned char hdr[14 + 108];
  std::memset(hdr, 0, sizeof(hdr));
  auto put32 = [&](int at, uint32_t v) { for (int k = 0; k < 4; ++k) hdr[at + k] = (unsigned char)(v >> (8 * k)); };
  auto put16 = [&](int at, uint32_t v) { hdr[at] = (unsigned char)v; hdr[at + 1] = (unsigned char)(v >> 8); };
  hdr[0] = 'B'; hdr[1] = 'M'; put32(2, off + data); put32(10, off);
  put32(14, 108); put32(18, w); put32(22, h); put16(26, 1); put16(28, 32); put32(30, 3); put32(34, data);
  put32(54, 0x00ff0000u); put32(58, 0x0000ff00u); put32(62, 0x000000ffu); put32(66, 0xff000000u);
  put32(70, 0x57696e20u);   // "Win " colour space
  FILE* f = std::fopen(filename, "wb");
  if (!f) throw std::runtime_error(std::string("cannot write ") + filename);
  std::fwrite(hdr, 1, sizeof(hdr), f);
  for (int y = s->height - 1; y >= 0; --y) std::fwrite(s->buffer + (size_t)y * s->width, 4, (size_t)s->width, f);
  std::fclose(f);
}

void KillSDL(screen* s) {
  delete[] s->buffer;
  delete s;
}

void DrawStandard(screen* scr, std::vector<PhotonBeam>, std::vector<PhotonBeam>) { render_frame(scr, PB_MODE_721); }

// BeamRadiance's per-segment dispatch (BeamRadiance.h:486-543): cones if ADAPTIVE, else by ONE_DIMENSIONAL
void Draw(screen* scr, std::vector<PhotonBeam>, std::vector<PhotonBeam>, std::vector<PhotonSeg>&) { render_frame(scr, PB_MODE_DEFAULT); }

void RunAnalysis() {
  pb_ctx* ctx = context();
  const double analysis_extinction = 0.04f;
  const char* names[2] = {"NSD.csv", "rRMSE.csv"};
  for (int table = 0; table < 2; ++table) {
    const int cols = table == 0 ? 8 : 3;
    int rows = 0;
    std::vector<double> first(1), values(cols);
    std::vector<uint8_t> present(cols);
    int rc = pb_run_analysis(ctx, analysis_extinction, table, first.data(), values.data(), present.data(), 0, &rows);
    if (rc != PB_OK && rc != PB_ERR_CAPACITY) check(rc, "pb_run_analysis");
    first.assign((size_t)rows + 1, 0.0); values.assign((size_t)rows * cols + 1, 0.0); present.assign((size_t)rows * cols + 1, 0);
    check(pb_run_analysis(ctx, analysis_extinction, table, first.data(), values.data(), present.data(), rows, &rows), "pb_run_analysis");
    std::ofstream file;
    file.open(names[table]);
    for (int r = 0; r < rows; ++r) {
      file << first[(size_t)r];                              // Analysis.h:93, :110
      for (int k = 0; k < cols; ++k) {
        file << ",";
        if (present[(size_t)r * cols + k]) file << std::to_string(values[(size_t)r * cols + k]);
      }
      file << "\n";
    }
    file.close();
  }
}

void DrawHet(screen* scr, std::vector<PhotonBeam>, std::vector<PhotonBeam>, NodeGen*) { render_frame(scr, PB_MODE_73HET); }

void LoadSegments(const char* path, const mat4& matrix, const std::vector<Triangle>& triangles) {
  extinction_c = absorption_c + scattering_c;
  recreate_context_if_control_changed();
  pb_ctx* ctx = context();
  std::vector<float> tri(triangles.size() * 9);
  for (size_t i = 0; i < triangles.size(); ++i) {
    const vec4* v[3] = {&triangles[i].v0, &triangles[i].v1, &triangles[i].v2};
    for (int k = 0; k < 3; ++k) { tri[i * 9 + k * 3] = v[k]->x; tri[i * 9 + k * 3 + 1] = v[k]->y; tri[i * 9 + k * 3 + 2] = v[k]->z; }
  }
  g_triangles = triangles;
  check(pb_set_scene(ctx, tri.data(), (int)triangles.size(), &matrix.m[0][0]), "pb_set_scene");
  remember_scene(tri, matrix);
  check(pb_load_segments(ctx, path), "pb_load_segments");
  if (devices.size() > 1) {      // the other devices get the set from the host
    int n = 0, got = 0;
    check(pb_segment_count(ctx, &n), "pb_segment_count");
    std::vector<pb_segment> segs((size_t)(n > 0 ? n : 1));
    check(pb_download_segments(ctx, segs.data(), (int)segs.size(), &got), "pb_download_segments");
    float lo[3], hi[3];
    check(pb_get_root_box(ctx, lo, hi), "pb_get_root_box");
    remember_render_set(segs.data(), (size_t)got, lo, hi);
  }
}
void SaveSegments(const char* path) { check(pb_save_segments(context(), path), "pb_save_segments"); }

// ---------------------------------------------------------------- the keyboard loop, without a window --
namespace { std::deque<int> g_events; constexpr int QUIT_EVENT = -1; }
void PushKey(int sdl_key_code) { g_events.push_back(sdl_key_code); }
void PushQuit() { g_events.push_back(QUIT_EVENT); }

bool UserInput() {
  const double PI = 3.14159265;                   // Tessendorf.h:4; the sums below are formed in double and rounded to float, as there
  while (!g_events.empty()) {
    const int key_code = g_events.front();
    g_events.pop_front();
    if (key_code == QUIT_EVENT) return false;
    switch (key_code) {
      // Rotation
      case SDLK_UP: theta.x += PI / 180; break;
      case SDLK_DOWN: theta.x -= PI / 180; break;
      case SDLK_LEFT: theta.y += PI / 180; break;
      case SDLK_RIGHT: theta.y -= PI / 180; break;
      // Move forward/back/left/right
      case SDLK_w: camera.z += 0.1; break;
      case SDLK_s: camera.z -= 0.1; break;
      case SDLK_a: camera.x -= 0.1; break;
      case SDLK_d: camera.x += 0.1; break;
      // Move the light
      case SDLK_l: light_position.x += 0.1; break;
      case SDLK_j: light_position.x -= 0.1; break;
      case SDLK_o: light_position.y -= 0.1; break;
      case SDLK_u: light_position.y += 0.1; break;
      case SDLK_k: light_position.z -= 0.1; break;
      case SDLK_i: light_position.z += 0.1; break;
      // Reset state
      case SDLK_r:
        camera = vec4{0, 0, -3.00f, 1};
        theta = vec3{0.0f, 0.0f, 0.0f};
        light_position = vec4{0, -0.5f, -0.7f, 1};
        break;
      case SDLK_ESCAPE: return false;
    }
  }
  return true;
}

bool Update() {
  using clock = std::chrono::steady_clock;
  static clock::time_point t = clock::now();       // SDL_GetTicks() at the first call
  const clock::time_point t2 = clock::now();
  const float dt = (float)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t).count();
  t = t2;
  std::cout << "Render time: " << dt << " ms." << std::endl;
  return UserInput();
}

pb_stats Stats() {
  pb_stats s;
  check(pb_counters(context(), &s), "pb_counters");
  return s;
}

void UseDevicesFromEnvironment() {
  const char* e = std::getenv("PB_DEVICES");
  if (!e || !*e) return;
  std::vector<int> list;
  for (const char* p = e; *p;) {
    char* end = nullptr;
    const long v = std::strtol(p, &end, 10);
    if (end == p || v < 0) throw std::runtime_error(std::string("PB_DEVICES: not a list of device indices: ") + e);
    list.push_back((int)v);
    p = (*end == ',') ? end + 1 : end;
    if (*end && *end != ',') throw std::runtime_error(std::string("PB_DEVICES: not a list of device indices: ") + e);
  }
  if (!list.empty()) devices = list;
}

void Shutdown() {
  drop_helpers();
  if (g_ctx) { pb_destroy(g_ctx); g_ctx = nullptr; }
}

}  // namespace pbhost
