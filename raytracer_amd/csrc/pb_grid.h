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

  // Heterogeneous.h:144-180 for one sample
  PB_DI float sample(const GridView& g, float dist, V3 index) const {
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
    const float left = tl + ((bl - tl) * prop_y);
    const float right = tr + ((br - tr) * prop_y);
    float ext = left + ((right - left) * prop_x);
    if (ext < 0) ext = 0;
    return ext;
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

// Return value of Extinction3D: the plain mean of all samples (Heterogeneous.h:182-184).
PB_DI float extinction3d_mean(const GridView& g, V3 start, V3 end) {
  DdaWalk w;
  w.begin(g, start, end);
  float sum = w.sample(g, 0.0f, w.idx);
  int n = 1;
  for (;;) {
    float t;
    const bool done = w.advance(t);
    const V3 at = done ? w.prev : w.idx;
    sum += w.sample(g, done ? w.t_max : t, at);
    n++;
    if (done) break;
    w.prev = w.idx;
  }
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
