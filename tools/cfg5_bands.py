#!/usr/bin/env python3
"""The largest configuration on one GPU: 4096^2 camera rays, 1 000 000 photons, BOUNCES 3, heterogeneous medium
(Integral_73Het), rendered as row bands so that no launch runs longer than a few seconds and progress is printed
per band.  A sparse lattice of pixels is checked against the CPU oracle's flat scan over the very same segments.
precise = 1 selects the exact-order estimator (one grid walk per valid pair; bar 1e-5), 0 the default
relaxed-arithmetic one (bar 2e-3, the frame tolerance).
    python tools/cfg5_bands.py [photons] [size] [band_rows] [lattice] [precise]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as pb  # noqa: E402

photons = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
band = int(sys.argv[3]) if len(sys.argv) > 3 else 512
lattice = int(sys.argv[4]) if len(sys.argv) > 4 else 16
precise = int(sys.argv[5]) if len(sys.argv) > 5 else 0
bar = 1e-5 if precise else 2e-3
budget_s = 240.0

kw = dict(photon_number=photons, bounces=3, screen_width=size, screen_height=size, heterogeneous=1, one_dimensional=1)
sigma, side = pb.make_grid()
ctx = pb.Context(pb.control_defaults(**kw))
ctx.set_scene()
ctx.set_grid(sigma, side)
ctx.set_precise(bool(precise))
nb, ns = ctx.cast_photon_beams()
ctx.build_tree()
st = ctx.counters()
print(f"{'exact-order' if precise else 'relaxed-arithmetic'} estimator; {photons} photons, BOUNCES 3, het: {nb} beams, {ns} segments; photon pass {st['ms_emit']:.2f} ms, "
      f"LBVH {st['ms_build']:.2f} ms", flush=True)

img = np.zeros((size, size, 3), np.float32)
t_dev = 0.0
t0 = time.perf_counter()
rows_done = 0
for y0 in range(0, size, band):
    h = min(band, size - y0)
    img[y0:y0 + h] = ctx.draw(0, y0, size, h, mode=pb.MODE_73HET, cull="bins")
    ms = ctx.counters()["ms_render"]
    t_dev += ms
    rows_done = y0 + h
    print(f"rows {y0}..{y0 + h - 1}: {ms:.1f} ms (max radiance {img[y0:y0 + h].max():.3f})", flush=True)
    if time.perf_counter() - t0 > budget_s:
        print(f"stopping after {rows_done} rows: wall budget of {budget_s:.0f} s reached", flush=True)
        break
rays = rows_done * size
print(f"frame part {rows_done}x{size}: {t_dev:.1f} ms on the device = {rays / t_dev / 1e3:.2f} Mray/s; "
      f"finite {bool(np.isfinite(img).all())}, max radiance {img.max():.3f}, lit pixels {(img[:rows_done] > 0).any(axis=2).mean():.3f}",
      flush=True)

# band boundaries must not show: a band that straddles the middle of the frame, rendered again with other cuts
y_mid = min(size // 2, rows_done) - band // 2
if y_mid >= 0 and y_mid + band <= rows_done:
    again = ctx.draw(0, y_mid, size, band, mode=pb.MODE_73HET, cull="bins")
    print(f"band at rows {y_mid}..{y_mid + band - 1} rendered on its own equals the frame's rows: "
          f"{bool(np.array_equal(again, img[y_mid:y_mid + band]))}", flush=True)

if lattice > 0:
    import orc  # noqa: E402  (the checker, never the thing measured)
    segs = ctx.download_segments()
    lo, hi = ctx.root_box()
    step = size // lattice
    ys = [y for y in range(step // 2, size, step) if y < rows_done][:lattice]
    tri, _ = orc.load_test_model()
    t1 = time.perf_counter()
    worst = 0.0
    n_checked = 0
    for x in range(step // 2, size, step):               # one column of the lattice per call: the oracle threads over rows
        ref, cnt = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(),
                                       segs, lo, hi, x, ys[0], 1, len(ys), step_x=step, step_y=step,
                                       mode=orc.MODE_73HET, grid_sigma=sigma, side=side)
        got = img[ys[0]::step, x][:len(ys)].astype(np.float64)
        ok = ref[:, 0] < 1000.0                           # below the fixed-point saturation of single contributions
        err = (np.abs(got - ref[:, 0]) / np.maximum(1.0, np.abs(ref[:, 0])))[ok]
        worst = max(worst, float(err.max()) if err.size else 0.0)
        n_checked += int(ok.sum())
        print(f"oracle column {x}: worst relative error so far {worst:.2e}, max reference radiance "
              f"{ref.max():.3f} ({time.perf_counter() - t1:.0f} s)", flush=True)
    print(f"lattice of {n_checked} pixel channels against the oracle: worst error {worst:.2e} "
          f"(bar {bar:g} of max(1, |ref|)): {'ok' if worst <= bar else 'FAIL'}", flush=True)
