#!/usr/bin/env python3
"""Headline benchmark of the photon-beam radiance path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2]

A step is one frame of the hot path (DrawStandard: per pixel ClosestIntersection + beam radiance
estimate) with the beam set resident in HBM.  With N > 1 (one process per GPU, launched by
torch.distributed.run) the 8-row strips of the image are interleaved over the ranks and the rendered
strips are all-gathered over RCCL inside the timed region; the beam set is emitted once on rank 0 and
broadcast before timing.  Rank 0 prints one JSON line.

Workloads (BASELINE.json `configs`):
  cfg1  512^2,  1000 photons, BOUNCES 0          (the reference's own CPU case)
  cfg2  512^2, 10000 photons, BOUNCES 3          (default: the metric's 512^2 configuration)
  cfg3  512^2, 10000 photons, BOUNCES 3, heterogeneous medium (Integral_73Het estimator)
  cfg4  1024^2, 100000 photons, BOUNCES 0
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfg1": dict(screen_width=512, screen_height=512, photon_number=1000, bounces=0),
    "cfg2": dict(screen_width=512, screen_height=512, photon_number=10000, bounces=3),
    "cfg3": dict(screen_width=512, screen_height=512, photon_number=10000, bounces=3, heterogeneous=1, one_dimensional=1),
    "cfg4": dict(screen_width=1024, screen_height=1024, photon_number=100000, bounces=0),
}
PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 peak (packed issue), MI355X_MICROARCH.md chip table


def flop_per_ray(n_cull, n_valid):
    """Algorithmic work the reference's semantics require per camera ray (FMA = 2): 30 triangles x 30,
    46 per pair that passes the gate boxes (slab test 20 + cylinder quadratic 26), 67 per valid hit."""
    return 900.0 + 46.0 * n_cull + 67.0 * n_valid


def cpu_baseline(args, control_kwargs, segs, root_lo, root_hi, grid=None):
    """The CPU restatement (oracle/, test infrastructure) of the same estimate on a bounded pixel
    lattice of the same frame, on this host's cores.  Reported beside the GPU number, never part of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    ctl = orc.control_defaults(**control_kwargs)
    tri, _ = orc.load_test_model()
    threads = min(orc.max_threads(), 16)
    w, h = control_kwargs["screen_width"], control_kwargs["screen_height"]
    # aim for ~10-20 s: the flat scan costs about 15 ns per (ray, segment) pair and core (box gates reject most)
    rays_budget = 15.0 * threads / (15e-9 * max(len(segs), 1))
    step = 1
    while (w // step) * (h // step) > rays_budget and step < 64:
        step *= 2
    nx, ny = w // step, h // step
    t0 = time.perf_counter()
    _, cnt = orc.render_segments(ctl, orc.tri_xyz(tri), orc.transformation_matrix(), segs, root_lo, root_hi,
                                 step // 2, step // 2, nx, ny, step_x=step, step_y=step, threads=threads,
                                 mode=orc.MODE_73HET if grid is not None else orc.MODE_721,
                                 grid_sigma=None if grid is None else grid[0], side=0 if grid is None else grid[1])
    dt = time.perf_counter() - t0
    return {"value": cnt["rays"] / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
            "sample": (f"every pixel of the same frame ({nx}x{ny} rays, {dt:.1f} s), " if step == 1 else
                       f"every {step}th pixel in x and y of the same frame ({nx}x{ny} rays, {dt:.1f} s), ") +
                      f"flat scan of all {len(segs)} segments per ray (the reference's BuildTree diverges "
                      f"above ~5200 photons, so its own tree cannot be built for this workload)",
            "n_cull_per_ray": cnt["cyl_calls"] / max(cnt["rays"], 1),
            "n_valid_per_ray": cnt["valid_hits"] / max(cnt["rays"], 1)}


def measured_traffic(workload, cull):
    """HBM bytes per frame from the committed PMC passes (rocprofv3 cannot run inside this process)."""
    path = os.path.join(ROOT, "profiles", f"r01_hbm_traffic_{workload}.json")
    try:
        d = json.load(open(path))
        return float(d["traffic_bytes"]) if d.get("cull") == cull else None
    except (OSError, ValueError, KeyError):
        return None


def pipelined_throughput(pb, torch, ctl, dev, segs, root, het, grid, mode, cull, W, H, in_flight, steps, warmup):
    """Same frames, `in_flight` of them at a time: one context (own workspace) and one stream per slot, so the
    set-up and binning launches of a frame run beside the evaluation kernel of the frames before it.  Every
    frame still does all of its work; reported beside `value`, which is measured without any overlap."""
    ctxs, streams, frames = [], [], []
    for _ in range(in_flight):
        c = pb.Context(ctl, device=dev.index)
        c.set_scene()
        if het:
            c.set_grid(*grid)
        c.upload_segments(segs, root[:3], root[3:])
        ctxs.append(c); streams.append(torch.cuda.Stream(device=dev))
        frames.append(torch.zeros((H, W, 3), dtype=torch.float32, device=dev))

    def run(k_frames):
        for k in range(k_frames):
            i = k % in_flight
            ctxs[i].draw_strips_device(frames[i].data_ptr(), 0, 1, (H + 7) // 8, mode=mode, cull=cull,
                                       stream=streams[i].cuda_stream)
    run(max(warmup, in_flight))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(bool(torch.equal(frames[0], f)) for f in frames[1:])
    for c in ctxs:
        c.close()
    rays = W * H * ctl.ssaa * ctl.ssaa
    return {"frames_in_flight": in_flight, "value": rays * steps / dt / 1e6, "unit": "Mray/s",
            "ms_per_step": dt / steps * 1e3, "frames_identical": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cull", default="bins", choices=["bins", "lbvh", "scan"],
                    help="how a tile finds its candidate segments (default: screen-space bins)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="frames kept in flight for the extra `pipelined` figure (0: skip it); `value` itself is always "
                         "measured one frame after the other on one stream")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as pb
    from raytracer_amd import tiles

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available() or pb.device_count() < 1:
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # a rank stuck in a collective would hold the whole job: say where it stands and end instead
        import faulthandler
        faulthandler.dump_traceback_later(600, exit=True)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    wk = WORKLOADS[args.workload]
    ctl = pb.control_defaults(**wk)
    W, H = wk["screen_width"], wk["screen_height"]
    ctx = pb.Context(ctl, device=local_rank)
    ctx.set_scene()
    het = bool(wk.get("heterogeneous"))
    mode = pb.MODE_73HET if het else pb.MODE_721
    grid = pb.make_grid() if het else None
    if het:
        ctx.set_grid(*grid)      # CreateSurface(27, -3.0, 0.016) moved into camera space (skeleton.cpp:113-125)

    # ---- beam set: emitted once on rank 0 (GPU photon pass), broadcast, resident everywhere ----
    if rank == 0:
        ctx.cast_photon_beams()
        ctx.build_tree()
        # again, with the buffers of the first pass in place: phases_ms is the warm photon pass (same seed, same beams)
        n_beams, n_segs = ctx.cast_photon_beams()
        segs = ctx.download_segments()
        root = np.concatenate(ctx.root_box()).astype(np.float32)
    if world > 1:
        buf, n_segs, root = tiles.broadcast_segments(segs if rank == 0 else None, root if rank == 0 else None, dev,
                                                     itemsize=pb.SEGMENT_DTYPE.itemsize)   # RCCL broadcast
        nb = torch.tensor([n_beams if rank == 0 else 0], device=dev)
        dist.broadcast(nb, 0)
        n_beams = int(nb.item())
        torch.cuda.synchronize()
        if rank != 0:
            ctx.upload_segments_device(buf.data_ptr(), n_segs, root[:3], root[3:],
                                       stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
    cull = args.cull
    ctx.build_tree()     # the LBVH is part of the photon pass whichever cull mode renders (ms_build)

    # ---- strips of this rank ----
    first_strip, strip_stride, n_my = tiles.strips_of_rank(H, rank, world)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    pipe = tiles.StripPipeline(world, n_my * 8, W, dev, frame=frame)   # the RCCL all-gather of frame k overlaps frame k + 1
    stream = torch.cuda.current_stream().cuda_stream
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        slot = pipe.begin(frame)
        if i is not None:
            ev0[i].record()
        ctx.draw_strips_device(pipe.mine[slot].data_ptr(), first_strip, strip_stride, n_my, mode=mode, cull=cull, stream=stream)
        if i is not None:
            ev1[i].record()
        pipe.submit(slot)

    for _ in range(args.warmup):
        step()
    pipe.drain(frame)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    pipe.drain(frame)              # every frame gathered and assembled inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    kt = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
    kernel_ms = float(kt.item())

    if rank == 0:
        rays = W * H * ctl.ssaa * ctl.ssaa
        value = rays * args.steps / dt / 1e6
        # algorithmic work per ray from the counters of one counted frame (not timed)
        ctx.enable_counters(True)
        full = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        ctx.draw_device(full.data_ptr(), 0, 0, W, H, mode=mode, cull=cull, stream=stream)
        torch.cuda.synchronize()
        cnt = ctx.counters()
        ctx.enable_counters(False)
        n_cull = cnt["gate_passes"] / rays
        n_valid = cnt["valid_hits"] / rays
        w_ray = flop_per_ray(n_cull, n_valid)
        rays_per_launch = rays / world
        achieved = rays_per_launch * w_ray / (kernel_ms * 1e-3) / 1e12
        frame_ok = bool(torch.isfinite(frame).all().item()) and float(frame.max().item()) > 0
        out = {
            "metric": "camera-ray x beam radiance queries/sec (Mray/s) at 512^2 Cornell box",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: Cornell box {W}x{H}, {wk['photon_number']} photons, "
                                   f"BOUNCES {wk['bounces']}, {'heterogeneous' if het else 'homogeneous'} medium, SEED 42 (GPU photon pass: "
                                   f"{n_beams} beams, {n_segs} segments, own-box gate rule)",
                       "rays_per_frame": rays, "segments": n_segs, "cull": cull,
                       "parallelism": f"row-strips x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": measured_traffic(args.workload, cull) if world == 1 else None,
                         "kernel": "frame = ray_setup + bin/cull + eval + finalize kernels", "kernel_ms": kernel_ms,
                         "flop_per_ray": w_ray, "n_cull_per_ray": n_cull, "n_valid_per_ray": n_valid,
                         "note": "FP32 vector-ALU bound (no MFMA, working set in L2); achieved = algorithmic "
                                 "flop per ray x rays per launch / measured launch time; traffic = HBM bytes per "
                                 "frame from separate rocprofv3 PMC passes (profiles/r01_hbm_traffic_*.json), "
                                 "mostly the pixel atomics, which execute at the memory side"},
            "pair_queries_per_s": rays * n_segs * args.steps / dt,
            "frame_ok": frame_ok,
            "per_ray": {k: cnt[k] / rays for k in ("candidates", "cyl_hits", "valid_hits", "gate_passes")},
            "phases_ms": {k: cnt[k] for k in ("ms_emit", "ms_prepare", "ms_build")},
            # one-time passes against HBM: algorithmic bytes (radix sort 6 passes x read+write x 12 B per segment,
            # boxes and nodes 128 B per segment; beams 80 B and segments 112 B written twice over) / device time
            "phases_gb_per_s": {
                "lbvh_build": (6 * 2 * 12 + 128) * n_segs / (cnt["ms_build"] * 1e-3) / 1e9 if cnt["ms_build"] > 0 else None,
                "photon_pass": (2 * 80 * n_beams + 2 * 112 * n_segs) / (cnt["ms_emit"] * 1e-3) / 1e9 if cnt["ms_emit"] > 0 else None},
        }
        if world == 1 and args.in_flight > 1:
            out["pipelined"] = pipelined_throughput(pb, torch, ctl, dev, segs, root, het, grid, mode, cull, W, H,
                                                    args.in_flight, args.steps, args.warmup)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, wk, segs, root[:3], root[3:], grid)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
    ctx.close()


if __name__ == "__main__":
    main()
