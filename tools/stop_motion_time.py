#!/usr/bin/env python3
"""Steady-state cost of one stop-motion frame (ProduceStopMotion, skeleton.cpp:265-328): new density field,
re-cast of the fixed beam list, LBVH, one heterogeneous frame.  Wall-clock per stage after a warm-up.
    python tools/stop_motion_time.py [photons] [size] [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as pb  # noqa: E402

photons = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = pb.Context(pb.control_defaults(photon_number=photons, screen_width=size, screen_height=size,
                                     heterogeneous=1, one_dimensional=1))
ctx.set_scene()
ctx.set_grid(*pb.make_grid(time=0.016))
ctx.cast_photon_beams()
ctx.build_tree()
ctx.draw(mode=pb.MODE_73HET, cull="bins")
acc = np.zeros(4)
for k in range(-3, frames):
    t0 = time.perf_counter()
    ctx.set_grid(*pb.make_grid(time=0.016 * (k + 5)))
    t1 = time.perf_counter()
    nb, ns = ctx.recast_photon_beams(seed=100 + k)
    t2 = time.perf_counter()
    ctx.build_tree()
    t3 = time.perf_counter()
    img = ctx.draw(mode=pb.MODE_73HET, cull="bins")
    t4 = time.perf_counter()
    if k >= 0:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
acc *= 1e3 / frames
print(f"{photons} photons, {size}^2, {frames} frames: grid {acc[0]:.2f} ms, recast {acc[1]:.2f} ms, LBVH {acc[2]:.2f} ms, "
      f"frame (with download) {acc[3]:.2f} ms; last frame: {nb} scattered beams, {ns} segments, max {img.max():.3f}")
