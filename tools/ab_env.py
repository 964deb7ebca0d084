"""A/B of an environment switch on one box: frames must be bit-identical, timings alternate.
   python tools/ab_env.py VAR[=value] [cfg1 cfg2 cfg4]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as pb
from bench import WORKLOADS

spec = sys.argv[1]
var, _, val = spec.partition("=")
val = val or "1"
names = sys.argv[2:] or ["cfg1", "cfg2", "cfg4"]
for name in names:
    kw = WORKLOADS[name]
    mode = pb.MODE_73HET if kw.get("heterogeneous") else pb.MODE_721
    with pb.Context(pb.control_defaults(**kw), device=0) as ctx:
        ctx.set_scene()
        if kw.get("heterogeneous"):
            ctx.set_grid(*pb.make_grid())
        ctx.cast_photon_beams()
        W, H = kw["screen_width"], kw["screen_height"]
        out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        frames, times = {}, {"off": [], "on": []}
        steps = {"cfg1": 400, "cfg2": 200}.get(name, 40)
        for rnd in range(3):
            for which in ("off", "on"):
                if which == "on": os.environ[var] = val
                else: os.environ.pop(var, None)
                pb.reload_tuning()
                for _ in range(5):
                    ctx.draw_device(out.data_ptr(), 0, 0, W, H, mode=mode, cull="bins", stream=stream)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(steps):
                    ctx.draw_device(out.data_ptr(), 0, 0, W, H, mode=mode, cull="bins", stream=stream)
                torch.cuda.synchronize()
                times[which].append((time.perf_counter() - t) / steps * 1e3)
                frames[which] = out.cpu().numpy().copy()
        os.environ.pop(var, None)
        same = np.array_equal(frames["off"], frames["on"])
        d = np.abs(frames["off"].astype(np.float64) - frames["on"]).max()
        print(f"{name} {spec}: off {min(times['off']):.4f} ms  on {min(times['on']):.4f} ms  identical={same} maxdiff={d:.3g} max={frames['off'].max():.4g}", flush=True)
