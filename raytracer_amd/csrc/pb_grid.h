// Extinction3D / FindAverageExtinction3D (Heterogeneous.h:29-185) on the device: the 3-D DDA through the
// (side-1) x (side-1) x 2 cell grid over [-1,1] x [-1,1] x [2,4] and the bilinear look-up of sigma_t at
// every cell crossing.  The reference collects the (distance, extinction) samples in a vector; here they
// are produced one at a time by DdaWalk::next() and consumed on the fly, in the same order and with the
// same float operations (translation units including this are built with -ffp-contract=off).
//
// The reference reads GRID.geometric_points out of bounds for samples whose cell index lies outside the
// grid (rays that start in front of it); such indices are clamped into the array here.
#ifndef PB_GRID_H
#define PB_GRID_H
#include "pb_device_math.h"

struct GridView {
  const float* z;   // GRID.geometric_points[i].z, i = row * side + column
  int side;         // GRID.side_points
  int fast_div;     // pb_set_grid found x / cell == fma(fma(-q, cell, x), r, q), q = x r, r = 1 / cell, for every x
  // Behind the field, 16-byte aligned: per point (column, row) the four values sample() reads for a cell index
  // inside the grid - {z[i], z[i + side] (z[i] in the last row), z[i + 1] (z[i] in the last column), the second
  // again} - so that a sample costs one 16-byte load instead of four scattered ones.  pb_set_grid writes both.
  PB_DI const float4* cells() const { return (const float4*)(z + (((size_t)side * side + 3) & ~(size_t)3)); }
};

struct DdaWalk {
  V3 o, dir, cell, delta, idx, prev;
  float tx, ty, tz, t_max, res_x, res_y, rcell;
  int state;        // 0: the sample at distance 0 is pending, 1: stepping, 2: finished
  int guard;

  PB_DI static void axis_setup(float d, float og, float c, float& delta, float& t) {
    if (d < 0) { delta = -c / d; t = (floorf(og / c) * c - og) / d; }
    else       { delta = c / d;  t = ((floorf(og / c) + 1) * c - og) / d; }
  }

  // Heterogeneous.h:34-79
  PB_DI void begin(const GridView& g, V3 start, V3 end) {
    const V3 grid_min = v3(-1, -1, 2);
    const V3 d = end - start;
    dir = normalize3(d);
    res_x = (float)(g.side - 1); res_y = (float)(g.side - 1);
    cell = v3(2.0f / res_x, 2.0f / res_y, 2.0f / 2.0f);
    rcell = 1.0f / cell.x;
    o = start - grid_min;
    t_max = length3(d);
    axis_setup(dir.x, o.x, cell.x, delta.x, tx);
    axis_setup(dir.y, o.y, cell.y, delta.y, ty);
    axis_setup(dir.z, o.z, cell.z, delta.z, tz);
    idx = v3(floorf(o.x / cell.x), floorf(o.y / cell.y), floorf(o.z / cell.z));
    prev = idx;
    state = 0;
    guard = 0;
  }

  // A sample whose four grid values have been requested and whose interpolation is still to do: fetch() then finish()
  // is sample(); a loop that fetches sample i + 1 before it finishes sample i keeps a load in flight instead of waiting
  // for each in turn.
  struct Fetched { float tl, bl, tr, br, prop_x, prop_y; };
  PB_DI static float finish(const Fetched& f) {
    const float left = f.tl + ((f.bl - f.tl) * f.prop_y);
    const float right = f.tr + ((f.br - f.tr) * f.prop_y);
    float ext = left + ((right - left) * f.prop_x);
    if (ext < 0) ext = 0;
    return ext;
  }
  PB_DI float sample(const GridView& g, float dist, V3 index) const { return finish(fetch(g, dist, index)); }

  // Heterogeneous.h:144-180 for one sample
  PB_DI Fetched fetch(const GridView& g, float dist, V3 index) const {
    const V3 p = o + (dir * dist);
    const V3 mn = v3(cell.x * index.x, cell.y * index.y, cell.z * index.z);
    const V3 dl = p - mn;
    // x / cell by two fused multiply-adds where pb_set_grid has shown them to round like the division (cell.x
    // and cell.y are the same number); an x below 2^-60 (other than zero), above 2^100 or not finite takes the division itself
    const float ax = fabsf(dl.x), ay = fabsf(dl.y);
    float prop_x, prop_y;
    const uint32_t bx = __float_as_uint(ax), by = __float_as_uint(ay);      // zero (common at a crossing) or [2^-60, 2^100)
    const bool in_range = (bx == 0u || bx - 0x21800000u < 0x50000000u) && (by == 0u || by - 0x21800000u < 0x50000000u);
    const float qx = ax * rcell, qy = ay * rcell;
    prop_x = __builtin_fmaf(__builtin_fmaf(-qx, cell.x, ax), rcell, qx);
    prop_y = __builtin_fmaf(__builtin_fmaf(-qy, cell.x, ay), rcell, qy);
    if (__builtin_expect(!(g.fast_div && in_range), 0)) {
      asm volatile("");      // an empty statement the compiler may not run early: both divisions stay behind the branch
      prop_x = ax / cell.x; prop_y = ay / cell.y;
    }
    const int side = g.side, last = side * side - 1;
    const int ix = (int)index.x, iy = (int)index.y;
    float tl, bl, tr, br;
    if ((unsigned)ix < (unsigned)side && (unsigned)iy < (unsigned)side) {
      // every index below stays inside the array, so the clamps are idle and the table holds the same four values
      const float4 c = g.cells()[iy * side + ix];
      tl = c.x; bl = c.y; tr = c.z; br = c.w;
    } else {
      int t_l = (int)((index.y * side) + index.x);
      int b_l = (int)(((index.y + 1) * side) + index.x);
      if (index.y + 1 >= side) b_l = t_l;
      int t_r = t_l + 1;
      if (index.x + 1 >= side) t_r = t_l;
      const int b_r = b_l;                     // Heterogeneous.h:158: the unbraced `b_r = b_l` always runs
      t_l = min(max(t_l, 0), last); b_l = min(max(b_l, 0), last); t_r = min(max(t_r, 0), last);
      tl = g.z[t_l]; bl = g.z[b_l]; tr = g.z[t_r]; br = g.z[min(max(b_r, 0), last)];
    }
    Fetched f;
    f.tl = tl; f.bl = bl; f.tr = tr; f.br = br; f.prop_x = prop_x; f.prop_y = prop_y;
    return f;
  }

  // One step of the walk (Heterogeneous.h:84-122): true when it leaves the grid or passes t_max, and the
  // closing sample (t_max, previous cell) is due; otherwise the crossing distance is in t and idx has moved.
  PB_DI bool advance(float& t) {
    // selects, not branches: with branches the compiler indexes idx as an array and puts it in scratch memory
    const bool step_x = (tx < ty) && (tx < tz);
    const bool step_y = !step_x && (ty < tz);
    const bool step_z = !step_x && !step_y;
    t = step_x ? tx : (step_y ? ty : tz);
    tx = step_x ? tx + delta.x : tx;
    ty = step_y ? ty + delta.y : ty;
    tz = step_z ? tz + delta.z : tz;
    idx.x = step_x ? idx.x + ((dir.x < 0) ? -1.0f : 1.0f) : idx.x;
    idx.y = step_y ? idx.y + ((dir.y < 0) ? -1.0f : 1.0f) : idx.y;
    idx.z = step_z ? idx.z + ((dir.z < 0) ? -1.0f : 1.0f) : idx.z;
    // a NaN direction (zero-length ray) never satisfies t > t_max: the guard ends such a walk
    return t > t_max || idx.x < 0 || idx.y < 0 || idx.x > res_x || idx.y > res_y || ++guard > 4096;
  }

  // Next (distance, extinction) sample; false once the list is exhausted.
  PB_DI bool next(const GridView& g, float& dist, float& ext) {
    if (state == 2) return false;
    float t = 0.0f;
    V3 at = idx;
    if (state == 0) state = 1;
    else if (advance(t)) { state = 2; t = t_max; at = prev; }
    else { prev = idx; at = idx; }
    dist = t;
    ext = sample(g, t, at);
    return true;
  }
};

// The walk of a view ray from the camera at (0, 0, 0), done once per ray instead of once per (ray, beam) pair.  The
// return value of Extinction3D for the ray cut at distance s is
//   (sample at 0 + the crossing samples with t <= s + the closing sample at s) / (their number),
// and which crossings there are, in which cells, does not depend on s.  view_walk_tabulate walks the ray once with
// DdaWalk (the reference's float operations, so that crossings of two axes at one lattice corner - common, since
// pixel directions have rational slopes - are taken in the reference's order and sampled in the reference's cells)
// and keeps, per axis, the running sums of its crossing samples.  view_walk_mean then needs the number of crossings
// of each axis up to s (they lie at regular distances t0 + j / rate), three table look-ups and the closing sample.
// The walk ends at the crossing that takes the x or y cell index outside the grid (Heterogeneous.h:117-120); no
// later crossing of any axis is sampled, which the per-axis limits `cap` reproduce.
constexpr int VIEW_WALK_XY = 16;     // table rows per lateral axis: running sums of 0..15 crossings (side <= 30)
constexpr int VIEW_WALK_Z = 8;       // and along z (cells are 1 deep: 7 crossings reach z = 5)
constexpr int VIEW_WALK_ROWS = 2 * VIEW_WALK_XY + VIEW_WALK_Z;
constexpr int VIEW_WALK_MAX_SIDE = 30;
constexpr int VIEW_WALK_PARAMS = 16;                                   // rows after the tables: ViewWalk's members, 16 floats per ray
constexpr int VIEW_WALK_STRIDE = VIEW_WALK_ROWS + VIEW_WALK_PARAMS;    // rows of 64 floats per ray group

struct ViewWalk {
  float t0[3], rate[3];   // distance of an axis' first crossing, crossings per unit distance
  float cap[3];           // crossings of the axis that the walk sampled
  float e0;               // the sample at distance 0
  float t_unsure;         // distance of the first crossing whose order the tabulated walk cannot vouch for
  float lx, ly;           // the ray's unit direction (x, y) and the length of its once-normalised vec4 direction, kept with
  float d1_len;           // the walk for the evaluation's second pass (set by view_walk_kernel, not by view_walk_tabulate)
};

// sigma_t at grid coordinates (px, py) = position - grid_min, read in cell (ix, iy) as DdaWalk::sample reads it.
PB_DI float grid_sigma_in_cell(const GridView& g, float px, float py, float ix, float iy, float cell, float rcell) {
  const float top = (float)(g.side - 1);
  const float cx = __builtin_amdgcn_fmed3f(ix, 0.0f, top), cy = __builtin_amdgcn_fmed3f(iy, 0.0f, top);
  const float prop_x = fabsf(px - cell * cx) * rcell, prop_y = fabsf(py - cell * cy) * rcell;
  const float4 c = g.cells()[(int)cy * g.side + (int)cx];
  const float left = c.x + ((c.y - c.x) * prop_y);
  const float right = c.z + ((c.w - c.z) * prop_y);
  return fmaxf(left + ((right - left) * prop_x), 0.0f);
}

// l: the ray's unit direction.  tab: VIEW_WALK_ROWS rows of 64 floats in LDS, this lane's column.
PB_DI ViewWalk view_walk_tabulate(const GridView& g, V3 l, float* tab, int lane) {
  ViewWalk w;
  DdaWalk d;
  d.begin(g, v3(0.0f, 0.0f, 0.0f), l * 64.0f);          // far beyond the scene: the walk ends by leaving the grid
  w.t0[0] = d.tx; w.t0[1] = d.ty; w.t0[2] = d.tz;
  w.rate[0] = 1.0f / d.delta.x; w.rate[1] = 1.0f / d.delta.y; w.rate[2] = 1.0f / d.delta.z;
  w.e0 = d.sample(g, 0.0f, d.idx);
  float n[3] = {0.0f, 0.0f, 0.0f}, run[3] = {0.0f, 0.0f, 0.0f};
  tab[0 * 64 + lane] = 0.0f; tab[VIEW_WALK_XY * 64 + lane] = 0.0f; tab[2 * VIEW_WALK_XY * 64 + lane] = 0.0f;
  // Two axes often cross at one lattice corner (pixel directions have rational slopes), and the cells in which the
  // two samples are read depend on which axis steps first.  The reference walks each pair's ray along
  // normalize(d1 * distance), which differs from l in the last bits from pair to pair: where the two crossing
  // distances are closer than that noise the order is the pair's own, and pairs cut beyond that point are left to
  // the exact-order code.  A tie of x and y on a diagonal ray is exact for every pair (y steps first).
  w.t_unsure = 3.0e38f;
  const bool diagonal = fabsf(l.x) == fabsf(l.y);
  bool alive = true;
  for (int step = 0; step < VIEW_WALK_ROWS - 3; ++step) {
    {
      const float first = fminf(d.tx, fminf(d.ty, d.tz)), second = __builtin_amdgcn_fmed3f(d.tx, d.ty, d.tz);
      const bool xy = (d.tz > second) || (d.tz > first && d.tx == d.ty);
      if (alive && (second - first) <= 1e-6f * second && !(diagonal && xy && d.tx == d.ty)) {
        // The order matters only through the cell in which the first of the two samples is read: the cell entered
        // by stepping one axis, or the other.  Where both read the same value at this point (all corners of a
        // cell but the one DdaWalk::sample reads from the wrong lattice point) either order gives the same mean.
        const float sgx = (d.dir.x < 0) ? -1.0f : 1.0f, sgy = (d.dir.y < 0) ? -1.0f : 1.0f;
        const bool x_in = d.tx <= second, y_in = d.ty <= second;      // the two axes of the tie (z cells do not enter a sample)
        V3 ia = d.idx, ib = d.idx;
        if (x_in && y_in) { ia.x += sgx; ib.y += sgy; }
        else if (x_in) ia.x += sgx;
        else ia.y += sgy;
        const float ea = d.sample(g, first, ia), eb = d.sample(g, first, ib);
        if (fabsf(ea - eb) > 1e-4f) w.t_unsure = fminf(w.t_unsure, first);
      }
    }
    const V3 before = d.idx;
    float t;
    const bool done = d.advance(t);
    alive = alive && !done;
    const int axis = (d.idx.x != before.x) ? 0 : ((d.idx.y != before.y) ? 1 : 2);
    const float limit = (axis == 2) ? (float)(VIEW_WALK_Z - 1) : (float)(VIEW_WALK_XY - 1);
    const float na = (axis == 0) ? n[0] : ((axis == 1) ? n[1] : n[2]);
    alive = alive && (na < limit);                      // a full table ends the tabulated part of the walk
    if (alive) {
      const float e = d.sample(g, t, d.idx);
      const int row = (axis == 0) ? 0 : ((axis == 1) ? VIEW_WALK_XY : 2 * VIEW_WALK_XY);
      const float sum = ((axis == 0) ? run[0] : ((axis == 1) ? run[1] : run[2])) + e;
      tab[(row + (int)na + 1) * 64 + lane] = sum;
      if (axis == 0) { run[0] = sum; n[0] += 1.0f; } else if (axis == 1) { run[1] = sum; n[1] += 1.0f; } else { run[2] = sum; n[2] += 1.0f; }
    }
  }
  w.cap[0] = n[0]; w.cap[1] = n[1]; w.cap[2] = n[2];
  return w;
}

// A ray's numbers sit together (four 16-byte accesses per ray): the evaluation's second pass reads them for the ray a
// queued pair belongs to, not for its own lane.
PB_DI void view_walk_store(const ViewWalk& w, float* rows, int lane) {
  float4* at = (float4*)rows + 4 * lane;
  at[0] = make_float4(w.t0[0], w.t0[1], w.t0[2], w.e0);
  at[1] = make_float4(w.rate[0], w.rate[1], w.rate[2], w.t_unsure);
  at[2] = make_float4(w.cap[0], w.cap[1], w.cap[2], 0.0f);
  at[3] = make_float4(w.lx, w.ly, w.d1_len, 0.0f);
}
PB_DI ViewWalk view_walk_unpack(float4 a, float4 b, float4 c, float4 d) {
  ViewWalk w;
  w.t0[0] = a.x; w.t0[1] = a.y; w.t0[2] = a.z; w.e0 = a.w;
  w.rate[0] = b.x; w.rate[1] = b.y; w.rate[2] = b.z; w.t_unsure = b.w;
  w.cap[0] = c.x; w.cap[1] = c.y; w.cap[2] = c.z;
  w.lx = d.x; w.ly = d.y; w.d1_len = d.z;
  return w;
}
PB_DI ViewWalk view_walk_load(const float* rows, int lane) {
  const float4* at = (const float4*)rows + 4 * lane;
  return view_walk_unpack(at[0], at[1], at[2], at[3]);
}

// Extinction3D's mean for the ray cut at distance s (the walk's t_max).
PB_DI float view_walk_mean(const GridView& g, const ViewWalk& w, V3 l, float s, const float* tab, int lane) {
  const float res = (float)(g.side - 1);
  const float cell = 2.0f / res, rcell = 1.0f / cell;
  float n[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float c = floorf((s - w.t0[k]) * w.rate[k]) + 1.0f;       // crossing j lies at t0 + j / rate
    n[k] = __builtin_amdgcn_fmed3f(c, 0.0f, w.cap[k]);              // NaN (an axis the ray runs along): none
  }
  // the closing sample: at s, in the cell the walk stands in (grid_sigma_in_cell, with its load issued beside the others)
  const float fx0 = floorf(1.0f / cell), fy0 = fx0;
  const float ix = fx0 + ((l.x < 0) ? -n[0] : n[0]), iy = fy0 + ((l.y < 0) ? -n[1] : n[1]);
  const float cx = __builtin_amdgcn_fmed3f(ix, 0.0f, res), cy = __builtin_amdgcn_fmed3f(iy, 0.0f, res);
  float4 c = g.cells()[(int)cy * g.side + (int)cx];
  float t0 = tab[(int)n[0] * 64 + lane], t1 = tab[(VIEW_WALK_XY + (int)n[1]) * 64 + lane],
        t2 = tab[(2 * VIEW_WALK_XY + (int)n[2]) * 64 + lane];
  // one wait for the four loads: left to itself the compiler requests each where the sum needs it, four trips to memory
  // one after the other
  asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
  // (the relaxed estimator's own arithmetic: fused, where Extinction3D's restatement rounds every product)
  const float px = __builtin_fmaf(l.x, s, 1.0f), py = __builtin_fmaf(l.y, s, 1.0f);
  const float prop_x = fabsf(__builtin_fmaf(-cell, cx, px)) * rcell, prop_y = fabsf(__builtin_fmaf(-cell, cy, py)) * rcell;
  const float left = __builtin_fmaf(c.y - c.x, prop_y, c.x);
  const float right = __builtin_fmaf(c.w - c.z, prop_y, c.z);
  const float e_end = fmaxf(__builtin_fmaf(right - left, prop_x, left), 0.0f);
  const float sum = w.e0 + t0 + t1 + t2 + e_end;
  return sum * __builtin_amdgcn_rcpf(n[0] + n[1] + n[2] + 2.0f);
}

// Return value of Extinction3D: the plain mean of all samples (Heterogeneous.h:182-184).
PB_DI float extinction3d_mean(const GridView& g, V3 start, V3 end) {
  DdaWalk w;
  w.begin(g, start, end);
  // the samples are added in the reference's order; the values of sample i + 1 are requested before sample i is
  // interpolated and added (where a step goes does not depend on what the grid holds)
  DdaWalk::Fetched pending = w.fetch(g, 0.0f, w.idx);
  float sum = 0.0f;
  int n = 1;
  for (;;) {
    float t;
    const bool done = w.advance(t);
    const V3 at = done ? w.prev : w.idx;
    const DdaWalk::Fetched cur = w.fetch(g, done ? w.t_max : t, at);
    sum += DdaWalk::finish(pending);
    pending = cur;
    n++;
    if (done) break;
    w.prev = w.idx;
  }
  sum += DdaWalk::finish(pending);
  return sum / (float)n;
}

// Running mean of the samples that lie before distance t_s (CastPhotons.h:437-443, :530-536).
PB_DI float extinction3d_mean_until(const GridView& g, V3 start, V3 end, float t_s) {
  DdaWalk w;
  w.begin(g, start, end);
  float mean = 0.0f, step = 0.0f, d, e;
  while (w.next(g, d, e)) {
    if (!(d < t_s)) break;
    mean = mean + ((e - mean) / (step + 1));
    step += 1.0f;
  }
  return mean;
}

#endif
