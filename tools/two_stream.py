"""Experiment: frames alternated over C contexts on C streams (frame k+1's set-up and binning run beside frame k's evaluation)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as pb

wk = dict(screen_width=512, screen_height=512, photon_number=10000, bounces=3)
dev = torch.device("cuda", 0)
for C in (1, 2, 3, 4):
    ctxs, streams, frames = [], [], []
    for c in range(C):
        ctx = pb.Context(pb.control_defaults(**wk)); ctx.set_scene(); ctx.cast_photon_beams(); ctx.build_tree()
        ctxs.append(ctx); streams.append(torch.cuda.Stream()); frames.append(torch.zeros((512, 512, 3), device=dev))
    def run(K):
        for k in range(K):
            c = k % C
            ctxs[c].draw_strips_device(frames[c].data_ptr(), 0, 1, 64, cull="bins", stream=streams[c].cuda_stream)
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(torch.equal(frames[0], f) for f in frames)
    print(f"contexts {C}: {dt / 200 * 1e3:.4f} ms/frame, {512 * 512 * 200 / dt / 1e6:.0f} Mray/s, frames equal {same}", flush=True)
    for ctx in ctxs: ctx.close()
