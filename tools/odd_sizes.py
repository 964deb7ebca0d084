"""Frames of awkward sizes: the three candidate searches must agree and match the oracle."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
import raytracer_amd as pb
tri, _ = orc.load_test_model()
for (w, h, ssaa) in [(37, 23, 3), (1, 1, 1), (8, 8, 1), (2048, 8, 1), (9, 1000, 1), (64, 64, 4)]:
    kw = dict(photon_number=400, bounces=1, screen_width=w, screen_height=h, ssaa=ssaa)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene(); ctx.cast_photon_beams(); segs = ctx.download_segments(); lo, hi = ctx.root_box(); ctx.build_tree()
        fr = {c: ctx.draw(cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
    sw, sh = min(w, 48), min(h, 48)
    ref, _ = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(), segs, lo, hi, 0, 0, sw, sh)
    e = {c: np.abs(fr[c][:sh, :sw] - ref).max() for c in fr}
    d = max(np.abs(fr["lbvh"] - fr["scan"]).max(), np.abs(fr["bins"] - fr["scan"]).max())
    print(f"{w}x{h} ssaa {ssaa}: max radiance {ref.max():.3f}, vs oracle {e}, between searches {d:.2e}", flush=True)
    assert d < 1e-5 and max(e.values()) < 2e-3 * max(1.0, ref.max())
print("ok")
