// Known-answer access to the device functions the frame and photon kernels are built from: one lane per
// query, host arrays in and out.  These entry points exist so that the very code the kernels inline can be
// checked against vectors produced by the reference's own functions; nothing on the render path calls them.
#include <vector>

#include "pb_internal.h"
#include "pb_pair.h"

namespace {

// The three forms of a camera direction the reference derives from `dir` (skeleton.cpp:345-349,
// Intersections.h:91, :313).
PB_DI Ray ray_from_dir(V4 dir, float maxd) {
  Ray r;
  r.l = normalize3(xyz(dir));
  r.dn = xyz(normalize4(dir));
  r.d1 = xyz(dir);
  r.maxd = maxd;
  r.active = true;
  return r;
}

__global__ __launch_bounds__(64) void probe_closest_kernel(const TriPrep* __restrict__ tris, int n_tri,
                                                           const float* __restrict__ start, const float* __restrict__ dir,
                                                           int n, unsigned char* __restrict__ found, float* __restrict__ pos,
                                                           float* __restrict__ dist, int* __restrict__ index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V4 s = v4(start[4 * i], start[4 * i + 1], start[4 * i + 2], start[4 * i + 3]);
  const V4 d = v4(dir[4 * i], dir[4 * i + 1], dir[4 * i + 2], dir[4 * i + 3]);
  V4 p = v4(0, 0, 0, 0);
  float t = 0.0f;
  int idx = -1;
  const bool hit = closest_intersection(tris, n_tri, s, d, p, t, idx);
  found[i] = hit ? 1 : 0;
  pos[4 * i] = hit ? p.x : 0.0f; pos[4 * i + 1] = hit ? p.y : 0.0f; pos[4 * i + 2] = hit ? p.z : 0.0f; pos[4 * i + 3] = hit ? p.w : 0.0f;
  dist[i] = hit ? t : 0.0f;
  index[i] = hit ? idx : -1;
}

// Candidate planes of HitBoundingBox for the origin (0,0,0); same rule as gate_setup in pb_prepare.hip.
PB_DI void probe_gate_setup(const float lo[3], const float hi[3], float cand[3], uint32_t& qflags) {
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

__global__ __launch_bounds__(64) void probe_box_kernel(const float* __restrict__ lo, const float* __restrict__ hi,
                                                       const float* __restrict__ dir, int n,
                                                       unsigned char* __restrict__ hit, float* __restrict__ distance) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float l[3] = {lo[3 * i], lo[3 * i + 1], lo[3 * i + 2]}, h[3] = {hi[3 * i], hi[3 * i + 1], hi[3 * i + 2]};
  float cand[3];
  uint32_t q;
  probe_gate_setup(l, h, cand, q);
  const V4 d = v4(dir[4 * i], dir[4 * i + 1], dir[4 * i + 2], dir[4 * i + 3]);
  float hd = -1.0f;
  const bool ok = box_gate(l, h, cand, q, xyz(normalize4(d)), 3.402823466e+38f, &hd);
  hit[i] = ok ? 1 : 0;
  distance[i] = ok ? hd : -1.0f;
}

// Pair i = (ray from the origin along dir[i], segment i).  out[i*12..]: HitCylinder's return value, valid,
// tb_minus, tb_plus, tc_minus, tc_plus, entry point, exit point (zeros where the reference leaves them
// unset); integral/status: the estimator's value for the pair and 0/1/2 as eval_pair_* return them.
__global__ __launch_bounds__(64) void probe_pair_kernel(const SegFrame* __restrict__ frames, const HetFrame* __restrict__ het,
                                                        const float4* __restrict__ capA, const float4* __restrict__ capB,
                                                        GridView grid, const float* __restrict__ dir, int n, int mode,
                                                        int one_dimensional, int fast, float extinction_c, float scattering_c,
                                                        float* __restrict__ out12, double* __restrict__ integral,
                                                        int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SegFrame f = frames[i];
  const Ray r = ray_from_dir(v4(dir[4 * i], dir[4 * i + 1], dir[4 * i + 2], dir[4 * i + 3]), 3.402823466e+38f);
  CylHit h;
  h.entry = v3(0, 0, 0); h.exit = v3(0, 0, 0); h.tb_minus = h.tb_plus = h.tc_minus = h.tc_plus = 0.0f;
  int res;
  if (f.qflags & 32u) {
    ConeFrame cf;
    __builtin_memcpy(&cf, &het[i], sizeof(cf));
    res = hit_cone_exact(f, cf, r, h);
  } else res = hit_cylinder_exact(f, r, h);
  float* o = out12 + (size_t)i * 12;
  for (int k = 0; k < 12; ++k) o[k] = 0.0f;
  o[0] = h.first_inte ? 1.0f : 0.0f;
  if (res == 2) {
    o[1] = 1.0f; o[2] = h.tb_minus; o[3] = h.tb_plus; o[4] = h.tc_minus; o[5] = h.tc_plus;
    o[6] = h.entry.x; o[7] = h.entry.y; o[8] = h.entry.z; o[9] = h.exit.x; o[10] = h.exit.y; o[11] = h.exit.z;
  }
  double val = 0.0;
  int st;
  unsigned long long wrong = 0;
  if (mode == PB_MODE_GENERIC) {
    const float4 A = capA[i], B = capB[i];
    st = eval_pair_generic(f, het[i], v3(A.x, A.y, A.z), v3(B.x, B.y, B.z), grid, one_dimensional != 0, r, extinction_c,
                           scattering_c, val);
  } else if (mode != PB_MODE_721) {
    const float4 A = capA[i], B = capB[i];
    st = eval_pair_73het(f, het[i], v3(A.x, A.y, A.z), v3(B.x, B.y, B.z), grid, r, extinction_c, scattering_c, val);
  } else if (fast) st = eval_pair_721_fast<false>(f, r, extinction_c, val, wrong);
  else st = eval_pair_721(f, r, extinction_c, val);
  integral[i] = (st == 2) ? val : 0.0;
  status[i] = st;
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  template <class T> T* as() { return (T*)p; }
};

}  // namespace

extern "C" int pb_probe_closest_intersection(pb_ctx* ctx, const float* start_xyzw, const float* dir_xyzw, int n,
                                             uint8_t* found, float* pos_xyzw, float* dist, int32_t* index) {
  if (!ctx || !start_xyzw || !dir_xyzw || !found || !pos_xyzw || !dist || !index || n < 0) { pb_set_error("pb_probe_closest_intersection: bad argument"); return PB_ERR_ARG; }
  if (!ctx->d_tris) { pb_set_error("pb_probe_closest_intersection: pb_set_scene has not been called"); return PB_ERR_STATE; }
  if (n == 0) return PB_OK;
  PB_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nn = (size_t)n;
  Buf b;
  PB_HIP(hipMalloc(&b.p, nn * (sizeof(float) * 13 + sizeof(int) + 4)));
  float* d_start = b.as<float>(); float* d_dir = d_start + 4 * nn; float* d_pos = d_dir + 4 * nn; float* d_dist = d_pos + 4 * nn;
  int* d_idx = (int*)(d_dist + nn); unsigned char* d_found = (unsigned char*)(d_idx + nn);
  PB_HIP(hipMemcpyAsync(d_start, start_xyzw, sizeof(float) * 4 * nn, hipMemcpyHostToDevice, s));
  PB_HIP(hipMemcpyAsync(d_dir, dir_xyzw, sizeof(float) * 4 * nn, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(probe_closest_kernel, dim3((n + 63) / 64), dim3(64), 0, s, ctx->d_tris, ctx->n_tri, d_start, d_dir, n,
                     d_found, d_pos, d_dist, d_idx);
  PB_HIP(hipGetLastError());
  PB_HIP(hipMemcpyAsync(found, d_found, nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(pos_xyzw, d_pos, sizeof(float) * 4 * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(dist, d_dist, sizeof(float) * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(index, d_idx, sizeof(int) * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  return PB_OK;
}

extern "C" int pb_probe_hit_bounding_box(pb_ctx* ctx, const float* lo_xyz, const float* hi_xyz, const float* dir_xyzw, int n,
                                         uint8_t* hit, float* entry_distance) {
  if (!ctx || !lo_xyz || !hi_xyz || !dir_xyzw || !hit || !entry_distance || n < 0) { pb_set_error("pb_probe_hit_bounding_box: bad argument"); return PB_ERR_ARG; }
  if (n == 0) return PB_OK;
  PB_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nn = (size_t)n;
  Buf b;
  PB_HIP(hipMalloc(&b.p, nn * (sizeof(float) * 11 + 4)));
  float* d_lo = b.as<float>(); float* d_hi = d_lo + 3 * nn; float* d_dir = d_hi + 3 * nn; float* d_dist = d_dir + 4 * nn;
  unsigned char* d_hit = (unsigned char*)(d_dist + nn);
  PB_HIP(hipMemcpyAsync(d_lo, lo_xyz, sizeof(float) * 3 * nn, hipMemcpyHostToDevice, s));
  PB_HIP(hipMemcpyAsync(d_hi, hi_xyz, sizeof(float) * 3 * nn, hipMemcpyHostToDevice, s));
  PB_HIP(hipMemcpyAsync(d_dir, dir_xyzw, sizeof(float) * 4 * nn, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(probe_box_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d_lo, d_hi, d_dir, n, d_hit, d_dist);
  PB_HIP(hipGetLastError());
  PB_HIP(hipMemcpyAsync(hit, d_hit, nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(entry_distance, d_dist, sizeof(float) * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  return PB_OK;
}

extern "C" int pb_probe_pairs(pb_ctx* ctx, const pb_segment* segs, const float* dir_xyzw, int n, int mode, int fast,
                              float* cyl_out12, double* integral, int32_t* status) {
  if (!ctx || !segs || !dir_xyzw || !cyl_out12 || !integral || !status || n < 0) { pb_set_error("pb_probe_pairs: bad argument"); return PB_ERR_ARG; }
  if (mode != PB_MODE_721 && mode != PB_MODE_73HET && mode != PB_MODE_73 && mode != PB_MODE_GENERIC) {
    pb_set_error("pb_probe_pairs: mode must be PB_MODE_721, PB_MODE_73, PB_MODE_73HET or PB_MODE_GENERIC"); return PB_ERR_ARG;
  }
  if (mode == PB_MODE_73HET && !ctx->d_grid) { pb_set_error("pb_probe_pairs: PB_MODE_73HET needs pb_set_grid"); return PB_ERR_STATE; }
  if (n == 0) return PB_OK;
  PB_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t nn = (size_t)n;
  Buf in, fr, out;
  PB_HIP(hipMalloc(&in.p, nn * (sizeof(pb_segment) + sizeof(float) * 4)));
  PB_HIP(hipMalloc(&fr.p, nn * (sizeof(SegFrame) + sizeof(HetFrame) + 2 * sizeof(float4)) + 256));
  PB_HIP(hipMalloc(&out.p, nn * (sizeof(double) + sizeof(float) * 12 + sizeof(int))));
  pb_segment* d_segs = in.as<pb_segment>(); float* d_dir = (float*)(d_segs + nn);
  SegFrame* d_frames = fr.as<SegFrame>(); float4* d_capA = (float4*)(d_frames + nn); float4* d_capB = d_capA + nn;
  HetFrame* d_het = (HetFrame*)(d_capB + nn);
  double* d_int = out.as<double>(); float* d_cyl = (float*)(d_int + nn); int* d_status = (int*)(d_cyl + 12 * nn);
  PB_HIP(hipMemcpyAsync(d_segs, segs, sizeof(pb_segment) * nn, hipMemcpyHostToDevice, s));
  PB_HIP(hipMemcpyAsync(d_dir, dir_xyzw, sizeof(float) * 4 * nn, hipMemcpyHostToDevice, s));
  int rc = pb_launch_prepare_segments(d_segs, n, mode, ctx->ctl.one_dimensional, ctx->ctl.scattering_c,
                                      ctx->ctl.absorption_c + ctx->ctl.scattering_c, d_frames, d_capA, d_capB, d_het, nullptr, s);
  if (rc != PB_OK) return rc;
  GridView g;
  const bool generic_het = mode == PB_MODE_GENERIC && ctx->ctl.one_dimensional && ctx->ctl.heterogeneous && ctx->d_grid;
  g.z = (mode == PB_MODE_73HET || generic_het) ? ctx->d_grid : nullptr; g.side = ctx->grid_side; g.fast_div = ctx->grid_fast_div;
  hipLaunchKernelGGL(probe_pair_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d_frames, d_het, d_capA, d_capB, g, d_dir, n, mode,
                     ctx->ctl.one_dimensional, fast, ctx->ctl.absorption_c + ctx->ctl.scattering_c, ctx->ctl.scattering_c, d_cyl, d_int, d_status);
  PB_HIP(hipGetLastError());
  PB_HIP(hipMemcpyAsync(cyl_out12, d_cyl, sizeof(float) * 12 * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(integral, d_int, sizeof(double) * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipMemcpyAsync(status, d_status, sizeof(int) * nn, hipMemcpyDeviceToHost, s));
  PB_HIP(hipStreamSynchronize(s));
  return PB_OK;
}
