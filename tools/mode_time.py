#!/usr/bin/env python3
"""Frame time of one estimator at 512^2 (binned mode): python tools/mode_time.py <adaptive 0|1> <one_dimensional 0|1> [photons]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as pb  # noqa: E402

ada, one_d = int(sys.argv[1]), int(sys.argv[2])
photons = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
ctx = pb.Context(pb.control_defaults(screen_width=512, screen_height=512, photon_number=photons, bounces=3,
                                     adaptive=ada, one_dimensional=one_d))
ctx.set_scene()
nb, ns = ctx.cast_photon_beams()
out = torch.zeros((512, 512, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ctx.draw_device(out.data_ptr(), 0, 0, 512, 512, mode=pb.MODE_DEFAULT, cull="bins", stream=stream)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    ctx.draw_device(out.data_ptr(), 0, 0, 512, 512, mode=pb.MODE_DEFAULT, cull="bins", stream=stream)
b.record()
torch.cuda.synchronize()
img = out.cpu().numpy()
print(f"adaptive {ada} one_dimensional {one_d}: {nb} beams, {ns} segments, {a.elapsed_time(b) / 10:.3f} ms per frame, "
      f"max {img.max():.3f}, finite {bool(np.isfinite(img).all())}")
