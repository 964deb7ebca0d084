"""md5 of a workload's frame (screen-space bins, third frame of the context): to compare two builds of the library.
   python tools/frame_hash.py cfg1 cfg2 ..."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as pb
from bench import WORKLOADS
for name in sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4"]:
    kw = WORKLOADS[name]
    het = bool(kw.get("heterogeneous"))
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        if het:
            ctx.set_grid(*pb.make_grid())
        ctx.cast_photon_beams()
        for _ in range(3):
            img = ctx.draw(mode=pb.MODE_73HET if het else pb.MODE_721, cull="bins")
        print(name, hashlib.md5(img.tobytes()).hexdigest(), float(img.max()), flush=True)
