import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import orc, raytracer_amd as pb
n_photons, bounces = 1500, 2
octl = orc.control_defaults(heterogeneous=1, one_dimensional=1, photon_number=n_photons, bounces=bounces)
o = orc.Oracle(octl); osigma, side = o.make_grid(); o.cast_photon_beams(rng_kind=1); o.bound_photon_beams()
ob = o.beams()
with pb.Context(pb.control_defaults(heterogeneous=1, one_dimensional=1, photon_number=n_photons, bounces=bounces)) as ctx:
    ctx.set_scene(); sigma, _ = pb.make_grid(); ctx.set_grid(sigma, side)
    ctx.cast_photon_beams(); gb = ctx.download_beams()
bad = np.flatnonzero((np.abs(gb["energy"] - ob["energy"]) > 2e-5 * np.abs(ob["energy"])).any(1))
print("bad energy beams", len(bad))
for i in bad:
    ph = gb["photon"][i]
    sel = np.flatnonzero(gb["photon"] == ph)
    print("photon", ph, "beam", i, "order", gb["order"][i], "flags", gb["flags"][i], "E gpu", gb["energy"][i][0], "E orc", ob["energy"][i][0])
    for j in sel:
        print("   order", gb["order"][j], "flags", gb["flags"][j], "hit", gb["index_hit"][j], "start_same", (gb["start"][j] == ob["start"][j]).all(), "end_same", (gb["end"][j] == ob["end"][j]).all(),
              "dstart", np.abs(gb["start"][j] - ob["start"][j]).max(), "dend", np.abs(gb["end"][j] - ob["end"][j]).max(), "E", gb["energy"][j][0], ob["energy"][j][0],
              "len", np.linalg.norm(gb["end"][j][:3]-gb["start"][j][:3]))
