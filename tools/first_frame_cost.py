"""What a frame of a new segment set costs (the sizing frame: every frame of a stop-motion loop is one) against a kept
frame of the same set.  Two segment sets of one workload are uploaded in turn from device memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as pb
from bench import WORKLOADS
for name in sys.argv[1:] or ["cfg1", "cfg2", "cfg3"]:
    kw = WORKLOADS[name]
    het = bool(kw.get("heterogeneous"))
    mode = pb.MODE_73HET if het else pb.MODE_721
    W, H = kw["screen_width"], kw["screen_height"]
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        if het:
            ctx.set_grid(*pb.make_grid())
        sets = []
        for seed in (1, 2):
            ctx.cast_photon_beams(seed=seed)
            segs = ctx.download_segments(); lo, hi = ctx.root_box()
            sets.append((torch.from_numpy(segs.view(np.uint8).copy()).cuda(), len(segs), lo, hi))
        out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        def loop(n, frames_per_set):
            torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(n):
                d, k, lo, hi = sets[i & 1]
                ctx.upload_segments_device(d.data_ptr(), k, lo, hi, stream=s)
                for _ in range(frames_per_set):
                    ctx.draw_device(out.data_ptr(), 0, 0, W, H, mode=mode, cull="bins", stream=s)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3
        loop(4, 1)
        up = loop(30, 0); one = loop(30, 1); two = loop(30, 2); three = loop(30, 3); many = loop(10, 13)
        print(f"{name}: upload + prepare {up:.3f} ms; first frame {one - up:.3f} ms; second (makes the kept queue) {two - one:.3f} ms; "
              f"third {three - two:.3f} ms; kept frames {(many - three) / 10:.3f} ms each", flush=True)
