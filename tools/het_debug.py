"""Relaxed-arithmetic against exact-order frames of the heterogeneous and the homogeneous 1-D estimators on a
256 x 256 frame of oracle-made segments: error quantiles and the pixels that differ most."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import orc
import raytracer_amd as pb
from test_gpu_het import het_control
n_photons, bounces = 600, 1
octl = het_control(orc, photon_number=n_photons, bounces=bounces, screen_width=256, screen_height=256)
o = orc.Oracle(octl)
sigma, side = o.make_grid()
o.cast_photon_beams(); o.bound_photon_beams()
segs = o.export_segments(tree=False)
lo, hi = o.root_box()
with pb.Context(het_control(pb, photon_number=n_photons, bounces=bounces, screen_width=256, screen_height=256)) as ctx:
    ctx.set_scene(); ctx.set_grid(sigma, side); ctx.upload_segments(segs, lo, hi); ctx.build_tree()
    fast = ctx.draw(mode=pb.MODE_73HET, cull="bins").astype(np.float64)
    ctx.set_precise(True)
    ex = ctx.draw(mode=pb.MODE_73HET, cull="bins").astype(np.float64)
    ctx.set_precise(False)
    # homogeneous Integral_73 for comparison (no grid)
    f73 = ctx.draw(mode=pb.MODE_73, cull="bins").astype(np.float64)
    ctx.set_precise(True)
    e73 = ctx.draw(mode=pb.MODE_73, cull="bins").astype(np.float64)
for name, a, b in (("73het", fast, ex), ("73", f73, e73)):
    d = a - b
    rel = np.abs(d) / np.maximum(np.abs(b), 1e-3)
    print(name, "max abs", np.abs(d).max(), "rmse", np.sqrt((d**2).mean()), "mean signed", d.mean(), "mean ref", b.mean())
    print("  rel quantiles", np.quantile(rel, [0.5, 0.9, 0.99, 0.999, 1.0]))
    idx = np.unravel_index(np.argsort(np.abs(d[..., 0]).ravel())[-8:], d.shape[:2])
    for y, x in zip(*idx): print("   px", y, x, "fast", a[y, x, 0], "exact", b[y, x, 0])
    print("  ratio of sums", a.sum() / b.sum())
