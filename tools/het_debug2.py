"""Finds, for the pixel where the relaxed-arithmetic and the exact-order heterogeneous frames differ most, the
segment responsible (bisection over the segment set) and prints it."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import raytracer_amd as pb
kw = dict(screen_width=512, screen_height=512, photon_number=10000, bounces=3, heterogeneous=1, one_dimensional=1)
sigma, side = pb.make_grid()
with pb.Context(pb.control_defaults(**kw)) as ctx:
    ctx.set_scene(); ctx.set_grid(sigma, side)
    ctx.cast_photon_beams()
    segs = ctx.download_segments(); lo, hi = ctx.root_box()
    def both(sel, x0=0, y0=0, w=512, h=512):
        ctx.upload_segments(sel, lo, hi)
        ctx.set_precise(False); a = ctx.draw(x0, y0, w, h, mode=pb.MODE_73HET, cull="bins").astype(np.float64)
        ctx.set_precise(True); b = ctx.draw(x0, y0, w, h, mode=pb.MODE_73HET, cull="bins").astype(np.float64)
        return a, b
    a, b = both(segs)
    d = np.abs(a - b)[..., 0] / np.maximum(1.0, b[..., 0])
    d[b[..., 0] > 1000] = 0
    print("pixels over the bar:", int((d > 2e-3).sum()), "max", d.max())
    order = np.argsort(d.ravel())[::-1][:5]
    for o in order:
        y, x = divmod(int(o), 512)
        print("px", y, x, "fast", a[y, x, 0], "exact", b[y, x, 0])
    y, x = divmod(int(order[0]), 512)
    x0, y0 = (x // 8) * 8, (y // 8) * 8
    cur = segs
    while len(cur) > 1:
        half = len(cur) // 2
        l, r = cur[:half], cur[half:]
        al, bl = both(l, x0, y0, 8, 8)
        dl = abs(al[y - y0, x - x0, 0] - bl[y - y0, x - x0, 0])
        ar, br = both(r, x0, y0, 8, 8)
        dr = abs(ar[y - y0, x - x0, 0] - br[y - y0, x - x0, 0])
        cur = l if dl >= dr else r
        print(len(cur), dl, dr)
    s = cur[0]
    al, bl = both(cur, x0, y0, 8, 8)
    print("segment", {k: s[k].tolist() for k in s.dtype.names})
    print("pixel", y, x, "fast", al[y - y0, x - x0], "exact", bl[y - y0, x - x0])
    np.save("gpurun_out/het_bad_segment.npy", cur)
