import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import raytracer_amd as pb
kw = dict(photon_number=600, bounces=1, screen_width=256, screen_height=256, heterogeneous=1, one_dimensional=1)
ctx = pb.Context(pb.control_defaults(**kw))
ctx.set_scene(); ctx.set_grid(*pb.make_grid()); ctx.cast_photon_beams()
def frames(env, n=3):
    os.environ.update(env); pb.reload_tuning()
    out = [ctx.draw(mode=pb.MODE_73HET, cull="bins") for _ in range(n)]
    for k in env: del os.environ[k]
    pb.reload_tuning()
    return out
ctx.set_precise(True)
exact = ctx.draw(mode=pb.MODE_73HET, cull="bins").astype(np.float64)
ctx.set_precise(False)
def rel(f):
    e = np.abs(f.astype(np.float64) - exact) / np.maximum(1.0, exact)
    return float(e.max()), int((e > 2e-3).sum())
base = frames({})
print("base vs exact", rel(base[0]))
print("base repeat", [np.array_equal(base[0], f) for f in base[1:]])
for env in (dict(PB_TUNE_WIDE="1"), dict(PB_TUNE_FUSED_MAX="0"), dict(PB_TUNE_WIDE="1", PB_TUNE_FUSED_MAX="0"), dict(PB_TUNE_NO_RESIDUE="1"), dict(PB_TUNE_HET1="1")):
    fs = frames(env)
    d = [np.abs(f.astype(np.float64) - base[0]).max() for f in fs]
    nd = [(f != base[0]).any(axis=2).sum() for f in fs]
    print(env, "maxdiff", d, "pixels differing", nd, "vs exact", rel(fs[0]))
