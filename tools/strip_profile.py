#!/usr/bin/env python3
"""Renders the strips of one rank of `world` (argv: world [rank] [frames]) on cfg2: run under rocprofv3 --kernel-trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as pb  # noqa: E402
from raytracer_amd import tiles  # noqa: E402

world = int(sys.argv[1]); rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0; frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = pb.Context(pb.control_defaults(screen_width=512, screen_height=512, photon_number=10000, bounces=3))
ctx.set_scene(); ctx.cast_photon_beams()
first, stride, n_my = tiles.strips_of_rank(512, rank, world)
out = torch.zeros((n_my * 8, 512, 3), dtype=torch.float32, device="cuda")
for _ in range(frames):
    ctx.draw_strips_device(out.data_ptr(), first, stride, n_my, cull="bins", stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
