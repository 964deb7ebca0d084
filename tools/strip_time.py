#!/usr/bin/env python3
"""Per-rank frame time of the row-strip sharding, rehearsed on one GPU: for world = 1, 2, 4, 8 renders the
strips of every rank in turn (cfg2) and prints the slowest rank's time: what an N-GPU run can reach at best
before the all-gather."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as pb  # noqa: E402
from raytracer_amd import tiles  # noqa: E402

kw = dict(screen_width=512, screen_height=512, photon_number=10000, bounces=3)
ctx = pb.Context(pb.control_defaults(**kw))
ctx.set_scene()
ctx.cast_photon_beams()
stream = torch.cuda.current_stream().cuda_stream
for world in [int(w) for w in os.environ.get("WORLDS", "1 2 4 8").split()]:
    worst = 0.0
    for rank in range(world):
        first, stride, n_my = tiles.strips_of_rank(512, rank, world)
        out = torch.zeros((n_my * 8, 512, 3), dtype=torch.float32, device="cuda")
        for _ in range(5):
            ctx.draw_strips_device(out.data_ptr(), first, stride, n_my, cull="bins", stream=stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            ctx.draw_strips_device(out.data_ptr(), first, stride, n_my, cull="bins", stream=stream)
        b.record()
        torch.cuda.synchronize()
        worst = max(worst, a.elapsed_time(b) / 30)
    print(f"world={world}: slowest rank {worst:.4f} ms per frame")
