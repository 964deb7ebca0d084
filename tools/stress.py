#!/usr/bin/env python3
"""Scale check on one GPU: photon pass, LBVH and one frame at sizes far above the bench workload.
    python tools/stress.py <photons> <bounces> <size> <het 0|1> [cull]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as pb  # noqa: E402

photons, bounces, size, het = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cull = sys.argv[5] if len(sys.argv) > 5 else "bins"
kw = dict(photon_number=photons, bounces=bounces, screen_width=size, screen_height=size)
if het:
    kw.update(heterogeneous=1, one_dimensional=1)
ctx = pb.Context(pb.control_defaults(**kw))
ctx.set_scene()
if het:
    ctx.set_grid(*pb.make_grid())
t0 = time.perf_counter()
nb, ns = ctx.cast_photon_beams()
t1 = time.perf_counter()
ctx.build_tree()
t2 = time.perf_counter()
mode = pb.MODE_73HET if het else pb.MODE_721
img = ctx.draw(mode=mode, cull=cull)
t3 = time.perf_counter()
img2 = ctx.draw(mode=mode, cull=cull)
t4 = time.perf_counter()
st = ctx.counters()
print(f"photons {photons} bounces {bounces} {size}^2 het {het}: {nb} beams, {ns} segments; host times: photon pass "
      f"{t1 - t0:.3f} s, LBVH {t2 - t1:.3f} s, first frame {t3 - t2:.3f} s, second frame {t4 - t3:.3f} s; device: emit "
      f"{st['ms_emit']:.2f} ms, build {st['ms_build']:.2f} ms, frame {st['ms_render']:.2f} ms; "
      f"max radiance {img.max():.3f}, finite {bool(np.isfinite(img).all())}, repeatable {bool(np.array_equal(img, img2))}")
