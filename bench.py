#!/usr/bin/env python3
"""Headline benchmark of the photon-beam radiance path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg1]

A step is one frame of the hot path (DrawStandard: per pixel ClosestIntersection + beam radiance
estimate) with the beam set resident in HBM.  With N > 1 (one process per GPU, launched by
torch.distributed.run) the 8-row strips of the image are interleaved over the ranks and the rendered
strips are all-gathered over RCCL inside the timed region; the beam set is emitted once on rank 0 and
broadcast before timing.  Rank 0 prints one JSON line.

Workloads (BASELINE.json `configs`):
  cfg1  512^2,  1000 photons, BOUNCES 0          (default: Control.h as committed, the configuration the
                                                  target is set on)
  cfg2  512^2, 10000 photons, BOUNCES 3
  cfg3  512^2, 10000 photons, BOUNCES 3, heterogeneous medium (Integral_73Het estimator)
  cfg4  1024^2, 100000 photons, BOUNCES 0
  cfg5  4096^2, 1000000 photons, BOUNCES 3, heterogeneous medium, in bands of 512 rows (under `other_workloads` on one GPU)
The line's `value`, `roofline` and `cpu_baseline` belong to --workload; `other_workloads` holds the same
measurement, shorter, for the rest (skipped with --only).
"""
import argparse
import datetime
import faulthandler
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
# BASELINE's fifth configuration (LBVH build / traversal stress): one GPU renders it as bands of rows, so that no launch
# runs for seconds; its line has no `roofline` (a counted frame would test 16.8 M rays against 23.7 M gate boxes)
BANDED = {"cfg5": dict(screen_width=4096, screen_height=4096, photon_number=1000000, bounces=3, heterogeneous=1,
                       one_dimensional=1, band_rows=512)}
OTHER_STEPS = {"cfg1": 50, "cfg2": 50, "cfg3": 10, "cfg4": 10, "cfg5": 2}
FRAMES_PER_GATHER = {"cfg1": 8, "cfg2": 2, "cfg3": 1, "cfg4": 1}     # jobs of several ranks: frames per strip all-gather
PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 peak: 256 CUs x 4 SIMDs x 32 lanes x 2 (FMA) x 2.4 GHz
STALL_SECONDS = 120        # a multi-rank run that stands still this long prints where and ends

# FP32 additions, multiplications, fused multiply-adds (2) and transcendentals of the fast pair evaluation
# (eval_pair_721_fast), by the stage a (ray, segment) pair reaches, and of the least a camera ray costs
# (normalisations + ClosestIntersection against the one triangle it hits).  Every operation counted here is
# executed, so the count is a lower bound of what the hardware counters see.
FLOP_PER_RAY = 80.0
FLOP_PER_CANDIDATE = 19.0      # R_inv rows x and z times l, a, b, discriminant
FLOP_PER_CYLINDER_HIT = 7.0    # square root, reciprocal, both roots
FLOP_PER_VALID_HIT = 38.0      # row y, caps, tc-/tc+, tb-, Integral_721


def algorithmic_flop(cnt, rays):
    """Lower bound of the FP32 work of one frame from its counters (one counted frame, not timed)."""
    return (FLOP_PER_RAY * rays + FLOP_PER_CANDIDATE * cnt["candidates"] + FLOP_PER_CYLINDER_HIT * cnt["cyl_hits"] +
            FLOP_PER_VALID_HIT * cnt["valid_hits"])


def reference_pairs_flop(cnt, rays):
    """What the same frame costs when every pair on which the reference calls HitCylinder is priced
    (900 per ray + 46 per pair behind its gate boxes + 67 per valid hit): the screen-space cull never
    evaluates most of those pairs, so this is an equivalent, not work done; kept as `pairs_equivalent`."""
    return 900.0 * rays + 46.0 * cnt["gate_passes"] + 67.0 * cnt["valid_hits"]


def committed_profile(workload, cull, stamp):
    """The rocprofv3 counter passes committed for this workload (tools/profile_frame.py; rocprofv3 cannot
    run inside this process), if they were measured on the kernel sources this process runs."""
    for rnd in ("r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{workload}.json")
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("cull") == cull:
            return d, os.path.relpath(path, ROOT), d.get("source_stamp") == stamp
    return None, None, False


def cpu_baseline(control_kwargs, segs, root_lo, root_hi, grid=None):
    """The CPU restatement (oracle/, test infrastructure) of the same estimate on a bounded sample of the
    same frame, on this host's cores.  Reported beside the GPU number, never part of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    ctl = orc.control_defaults(**control_kwargs)
    tri, _ = orc.load_test_model()
    threads = min(orc.max_threads(), 16)
    w, h = control_kwargs["screen_width"], control_kwargs["screen_height"]
    # aim for 10-20 s: the flat scan costs about 15 ns per (ray, segment) pair and core (box gates reject most)
    rays_budget = 15.0 * threads / (15e-9 * max(len(segs), 1))
    step = 1
    while (w // step) * (h // step) > rays_budget and step < 64:
        step *= 2
    nx, ny = w // step, h // step
    rays, frames, cyl, valid = 0, 0, 0, 0
    t0 = time.perf_counter()
    while True:
        _, cnt = orc.render_segments(ctl, orc.tri_xyz(tri), orc.transformation_matrix(), segs, root_lo, root_hi,
                                     step // 2, step // 2, nx, ny, step_x=step, step_y=step, threads=threads,
                                     mode=orc.MODE_73HET if grid is not None else orc.MODE_721,
                                     grid_sigma=None if grid is None else grid[0], side=0 if grid is None else grid[1])
        rays += cnt["rays"]; cyl += cnt["cyl_calls"]; valid += cnt["valid_hits"]; frames += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or frames >= 200:
            break
    what = "every pixel" if step == 1 else f"every {step}th pixel in x and y"
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
            "sample": f"{what} of the same frame ({nx}x{ny} rays), {frames} time(s) over, {dt:.1f} s: flat scan of all "
                      f"{len(segs)} segments per ray behind the reference's box gates (its BuildTree drops segments "
                      f"and diverges above ~5200 photons, so its own tree is not the structure timed here)",
            "n_cull_per_ray": cyl / max(rays, 1), "n_valid_per_ray": valid / max(rays, 1)}


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


class Bench:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import raytracer_amd as pb
        from raytracer_amd import tiles
        self.torch, self.dist, self.pb, self.tiles, self.args = torch, dist, pb, tiles, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            if self.rank == 0:
                print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; launch with torch.distributed.run", file=sys.stderr)
            sys.exit(2)
        if not torch.cuda.is_available() or pb.device_count() < 1:
            print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        # --rehearse-exchange: a job of one rank goes through everything a job of several does (RCCL process group,
        # segment broadcast, pipelined strip gather on the collective's stream, watchdog), so that code has run on
        # the hardware before a node with several GPUs sees it
        self.exchange = self.world > 1 or args.rehearse_exchange
        if self.exchange:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            self.watchdog()
            # a collective that never completes ends the job after STALL_SECONDS with RCCL's own report
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev,
                                    timeout=datetime.timedelta(seconds=STALL_SECONDS))

    def watchdog(self, i=None):
        """(Re-)armed as the steps go by (every 16th: arming costs the host several microseconds of a 40-microsecond
        step): STALL_SECONDS without progress ends the process with every thread's stack on stderr and exit status 1,
        instead of holding the whole job in a collective."""
        if self.exchange and (i is None or i % 16 == 0):
            faulthandler.dump_traceback_later(STALL_SECONDS, exit=True)

    def close(self):
        if self.exchange:
            self.dist.barrier()
            self.dist.destroy_process_group()
            faulthandler.cancel_dump_traceback_later()

    def run_banded(self, name, steps, warmup):
        """A frame too long for one launch, on one GPU: photon pass + LBVH (once), then `warmup` + `steps` frames, each a
        sequence of row bands through pb_render_d's pixel rectangle into one resident image."""
        torch, pb, dev = self.torch, self.pb, self.dev
        wk = dict(BANDED[name])
        band = wk.pop("band_rows")
        ctl = pb.control_defaults(**wk)
        W, H = wk["screen_width"], wk["screen_height"]
        ctx = pb.Context(ctl, device=self.local_rank)
        ctx.set_scene()
        grid = pb.make_grid()
        ctx.set_grid(*grid)
        ctx.cast_photon_beams()
        ctx.build_tree()
        n_beams, n_segs = ctx.cast_photon_beams()      # warm: the buffers of the first pass are in place
        t_build = time.perf_counter()
        ctx.build_tree()
        torch.cuda.synchronize()
        build_wall_ms = (time.perf_counter() - t_build) * 1e3
        st = ctx.counters()
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def one_frame():
            for y0 in range(0, H, band):
                h = min(band, H - y0)
                ctx.draw_device(frame[y0:y0 + h].data_ptr(), 0, y0, W, h, mode=pb.MODE_73HET, cull="bins", stream=stream)
        for _ in range(warmup):
            one_frame()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            one_frame()
        ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = W * H
        frame_ok = bool(torch.isfinite(frame).all().item()) and float(frame.max().item()) > 0
        out = {"value": rays * steps / dt / 1e6, "unit": "Mray/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
               "config": {"workload": f"{name}: Cornell box {W}x{H}, {wk['photon_number']} photons, BOUNCES {wk['bounces']}, heterogeneous "
                                      f"medium, SEED 42 (GPU photon pass: {n_beams} beams, {n_segs} segments, own-box gate rule), "
                                      f"bands of {band} rows, every band binned anew in every frame",
                          "rays_per_frame": rays, "segments": n_segs, "cull": "bins", "parallelism": "single GPU"},
               "kernel_ms": ev0.elapsed_time(ev1) / steps, "frame_ok": frame_ok,
               "lit_fraction": float((frame > 0).any(dim=2).float().mean().item()),
               "phases_ms": {"ms_emit": st["ms_emit"], "ms_prepare": st["ms_prepare"], "ms_build": st["ms_build"],
                             "build_wall_ms": build_wall_ms},
               "phases_gb_per_s": {
                   "lbvh_build": (6 * 2 * 12 + 128) * n_segs / (st["ms_build"] * 1e-3) / 1e9 if st["ms_build"] > 0 else None,
                   "photon_pass": (2 * 80 * n_beams + 2 * 112 * n_segs) / (st["ms_emit"] * 1e-3) / 1e9 if st["ms_emit"] > 0 else None}}
        ctx.close()
        del frame
        torch.cuda.empty_cache()
        return out

    def run(self, name, steps, warmup, with_extras):
        """One workload: photon pass + LBVH (once, untimed), `warmup` + `steps` frames, the counted frame.
        Returns the JSON fields of this workload on rank 0, None elsewhere."""
        torch, dist, pb, tiles, args = self.torch, self.dist, self.pb, self.tiles, self.args
        rank, world, dev = self.rank, self.world, self.dev
        wk = WORKLOADS[name]
        ctl = pb.control_defaults(**wk)
        W, H = wk["screen_width"], wk["screen_height"]
        ctx = pb.Context(ctl, device=self.local_rank)
        ctx.set_scene()
        het = bool(wk.get("heterogeneous"))
        mode = pb.MODE_73HET if het else pb.MODE_721
        grid = pb.make_grid() if het else None
        if het:
            ctx.set_grid(*grid)      # CreateSurface(27, -3.0, 0.016) moved into camera space (skeleton.cpp:113-125)

        # ---- beam set: emitted once on rank 0 (GPU photon pass), broadcast, resident everywhere ----
        segs = root = None
        n_beams = n_segs = 0
        if rank == 0:
            ctx.cast_photon_beams()
            ctx.build_tree()
            # again, with the buffers of the first pass in place: phases_ms is the warm photon pass (same seed, same beams)
            n_beams, n_segs = ctx.cast_photon_beams()
            segs = ctx.download_segments()
            root = np.concatenate(ctx.root_box()).astype(np.float32)
        if self.exchange:
            buf, n_segs, root = tiles.broadcast_segments(segs, root, dev, itemsize=pb.SEGMENT_DTYPE.itemsize)   # RCCL broadcast
            nb = torch.tensor([n_beams if rank == 0 else 0], device=dev)
            dist.broadcast(nb, 0)
            n_beams = int(nb.item())
            torch.cuda.synchronize()
            if rank != 0 or args.rehearse_exchange:     # the rehearsal renders from the broadcast records, as ranks > 0 do
                ctx.upload_segments_device(buf.data_ptr(), n_segs, root[:3], root[3:],
                                           stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
        cull = args.cull
        t_build = time.perf_counter()
        ctx.build_tree()     # the LBVH is part of the photon pass whichever cull mode renders (ms_build)
        torch.cuda.synchronize()
        build_wall_ms = (time.perf_counter() - t_build) * 1e3

        # ---- strips of this rank ----
        first_strip, strip_stride, n_my = tiles.strips_of_rank(H, rank, world)
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        # frames whose strips share one collective: a 512^2 frame is shorter than a collective's enqueue on the host
        fpg = args.frames_per_gather if args.frames_per_gather > 0 else (FRAMES_PER_GATHER.get(name, 1) if self.exchange else 1)
        pipe = tiles.StripPipeline(world, n_my * 8, W, dev, frame=frame, timeout_s=STALL_SECONDS, exchange=self.exchange,
                                   frames_per_gather=fpg)
        fpg = pipe.K
        stream = torch.cuda.current_stream().cuda_stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def render(mine, i):
            ctx.draw_strips_device(mine.data_ptr(), first_strip, strip_stride, n_my, mode=mode, cull=cull, stream=stream)

        tiles.run_steps(pipe, frame, render, warmup, on_step=self.watchdog)
        torch.cuda.synchronize()
        if self.exchange:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # one pair of HIP events round the whole timed region, on the stream the frames are launched on: per-frame
        # pairs cost a 0.06 ms frame a tenth of its time (each record is a barrier packet between its kernels)
        ev0.record()
        tiles.run_steps(pipe, frame, render, steps, on_step=self.watchdog)   # every frame gathered and assembled in here
        ev1.record()
        torch.cuda.synchronize()
        if self.exchange:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if self.exchange:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        kernel_ms = ev0.elapsed_time(ev1) / steps
        # How much one timed region of `steps` frames scatters on this box: the same region four more times (not part
        # of `value`, which is the region above and nothing else)
        repeats = []
        if with_extras and dt < 2.0:
            for _ in range(4):
                if self.exchange:
                    dist.barrier()
                torch.cuda.synchronize()
                tr = time.perf_counter()
                tiles.run_steps(pipe, frame, render, steps, on_step=self.watchdog)
                torch.cuda.synchronize()
                repeats.append((time.perf_counter() - tr) / steps * 1e3)
        # The same frames with nothing kept from one to the next: counting pass, queue build, fill, evaluation and
        # pixel write-out every frame (the lists, their offsets and the item queue are functions of the segment set and
        # the frame geometry alone, and `value` is measured with them kept from the first frame, as a tree would be).
        unkept_ms = None
        if world == 1 and with_extras and not self.exchange:
            os.environ["PB_TUNE_NO_KEEP"] = "1"
            pb.reload_tuning()
            try:
                tiles.run_steps(pipe, frame, render, max(2, warmup // 2))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                tiles.run_steps(pipe, frame, render, max(10, steps // 2))
                torch.cuda.synchronize()
                unkept_ms = (time.perf_counter() - t1) / max(10, steps // 2) * 1e3
            finally:
                del os.environ["PB_TUNE_NO_KEEP"]
                pb.reload_tuning()
            tiles.run_steps(pipe, frame, render, 2)      # the kept queue is in place again for what follows
            torch.cuda.synchronize()
        kt = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
        if self.exchange:
            dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kernel_ms = float(kt.item())
        self.watchdog()

        out = None
        if rank == 0:
            rays = W * H * ctl.ssaa * ctl.ssaa
            # work per frame from the counters of one counted frame (not timed)
            ctx.enable_counters(True)
            full = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            ctx.draw_device(full.data_ptr(), 0, 0, W, H, mode=mode, cull=cull, stream=stream)
            torch.cuda.synchronize()
            cnt = ctx.counters()
            ctx.enable_counters(False)
            stamp = pb.source_stamp()
            prof, prof_path, prof_fresh = committed_profile(name, cull, stamp)
            alg = algorithmic_flop(cnt, rays)
            use_hw = bool(prof and prof_fresh and prof.get("fp32_flop_per_frame") and world == 1)
            # Issued: 64 x the FP32 wave instructions the hardware counted.  Active: the same weighted, kernel by kernel,
            # with the share of lanes the execution mask left on (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU): a wave
            # instruction of which half the lanes are masked off does half the work.  `achieved` is the active figure.
            issued_frame = float(prof["fp32_flop_per_frame"]) if use_hw else None
            active_frame = float(prof["fp32_flop_active_lanes_per_frame"]) if (use_hw and prof.get("fp32_flop_active_lanes_per_frame")) else None
            flop_frame = active_frame if active_frame else (issued_frame if issued_frame else alg)
            # one launch = this rank's share of the frame (the strips are interleaved, so 1 / world of the work)
            achieved = flop_frame / world / (kernel_ms * 1e-3) / 1e12
            frame_ok = bool(torch.isfinite(frame).all().item()) and float(frame.max().item()) > 0
            out = {
                "value": rays * steps / dt / 1e6, "unit": "Mray/s", "steps": steps, "warmup": warmup,
                "ms_per_step": dt / steps * 1e3,
                "repeats_ms_per_step": repeats if repeats else None,
                "config": {"workload": f"{name}: Cornell box {W}x{H}, {wk['photon_number']} photons, BOUNCES {wk['bounces']}, "
                                       f"{'heterogeneous' if het else 'homogeneous'} medium, SEED 42 (GPU photon pass: "
                                       f"{n_beams} beams, {n_segs} segments, own-box gate rule); primary rays and the radiance "
                                       f"estimate every frame, with the binning schedule of the segment set (lists, survivors per "
                                       f"tile, item queue) kept from its first frames as a tree would be",
                           "rays_per_frame": rays, "segments": n_segs, "cull": cull,
                           "frames_per_gather": fpg if self.exchange else None,
                           "parallelism": f"row-strips x{world}" if world > 1 else
                                          ("single GPU, strips exchanged over RCCL as with several" if self.exchange else "single GPU")},
                "roofline": {
                    "bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_TFLOPS,
                    "traffic": float(prof["traffic_bytes"]) if (use_hw and prof.get("traffic_bytes")) else None,
                    "kernel": "the frame's launches together: the kept-schedule frame is ray set-up then evaluation with the pixel "
                              "write-out in it, or the evaluation alone where its items set up their own rays (cfg1)",
                    "kernel_ms": kernel_ms,
                    "kernel_ms_is": "HIP-event time of the timed region / steps: launches of a frame and the gaps between them",
                    "flop_per_frame": flop_frame,
                    "issued": None if issued_frame is None else {
                        "flop_per_frame": issued_frame, "tflops": issued_frame / world / (kernel_ms * 1e-3) / 1e12,
                        "frac": issued_frame / world / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                        "note": "64 lanes x FP32 wave instructions, masked lanes included"},
                    "lane_active_frac": (active_frame / issued_frame) if (active_frame and issued_frame) else None,
                    "flop_source": (f"hardware-counted: 64 x (ADD + MUL + 2 FMA + TRANS) FP32 wave instructions of the frame's kernels"
                                    f"{', each kernel weighted with its share of active lanes (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU)' if active_frame else ''}, "
                                    f"{prof_path} (measured on these kernel sources, stamp {stamp})") if use_hw else
                                   ("counted from this run's pairs: %g per ray + %g per evaluated pair + %g per cylinder hit + %g per "
                                    "valid hit (operations the kernels execute; a lower bound of the hardware count%s)" %
                                    (FLOP_PER_RAY, FLOP_PER_CANDIDATE, FLOP_PER_CYLINDER_HIT, FLOP_PER_VALID_HIT,
                                     "" if prof is None else f"; {prof_path} is from other kernel sources" if not prof_fresh else "")),
                    "algorithmic_flop_per_frame": alg,
                    "pairs_equivalent": {"flop_per_frame": reference_pairs_flop(cnt, rays),
                                         "tflops": reference_pairs_flop(cnt, rays) / world / (kernel_ms * 1e-3) / 1e12,
                                         "note": "900 per ray + 46 per pair behind the reference's gate boxes + 67 per valid "
                                                 "hit: what a traversal that calls HitCylinder on every such pair would "
                                                 "execute; not work these kernels do, not a fraction of the peak"},
                    "note": "FP32 vector-ALU bound (no MFMA; the working set is cache-resident, traffic = HBM bytes per frame "
                            "from the FETCH_SIZE / WRITE_SIZE passes of the same profile); achieved = FP32 operations per "
                            "launch / launch time measured here with HIP events on the launch stream"},
                "schedule_rebuilt_every_frame": None if unkept_ms is None else {
                    "ms_per_step": unkept_ms, "value": rays / (unkept_ms * 1e-3) / 1e6, "unit": "Mray/s",
                    "note": "the same frames with both binning passes and the queue build repeated every frame "
                            "(five launches instead of two); `value` keeps the lists, their offsets and the item queue, "
                            "which depend on the segment set and the frame geometry only, from the first frame of a "
                            "segment set (as a tree would be kept)"},
                "pair_queries_per_s": rays * n_segs * steps / dt,
                "frame_ok": frame_ok,
                "per_ray": {k: cnt[k] / rays for k in ("candidates", "cyl_hits", "valid_hits", "gate_passes")},
                "phases_ms": {"ms_emit": cnt["ms_emit"], "ms_prepare": cnt["ms_prepare"], "ms_build": cnt["ms_build"],
                              "build_wall_ms": build_wall_ms},
                # one-time passes against HBM: algorithmic bytes (radix sort 6 passes x read+write x 12 B per segment,
                # boxes and nodes 128 B per segment; beams 80 B and segments 112 B written twice over) / device time
                "phases_gb_per_s": {
                    "lbvh_build": (6 * 2 * 12 + 128) * n_segs / (cnt["ms_build"] * 1e-3) / 1e9 if cnt["ms_build"] > 0 else None,
                    "photon_pass": (2 * 80 * n_beams + 2 * 112 * n_segs) / (cnt["ms_emit"] * 1e-3) / 1e9 if cnt["ms_emit"] > 0 else None},
            }
            if with_extras:
                if world == 1 and args.in_flight > 1:
                    out["pipelined"] = pipelined_throughput(pb, torch, ctl, dev, segs, root, het, grid, mode, cull, W, H,
                                                            args.in_flight, steps, warmup)
                out["cpu_baseline"] = (cpu_baseline(wk, segs, root[:3], root[3:], grid)
                                       if (not args.no_cpu_baseline and world == 1) else None)
        ctx.close()
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg1", choices=sorted(WORKLOADS) + sorted(BANDED))
    ap.add_argument("--cull", default="bins", choices=["bins", "lbvh", "scan"],
                    help="how a tile finds its candidate segments (default: screen-space bins)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="measure --workload alone (no `other_workloads`)")
    ap.add_argument("--no-banded", action="store_true", help="leave the banded workload (cfg5, about 15 s) out of `other_workloads`")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="one rank, but through the RCCL process group, the segment broadcast and the pipelined strip "
                         "gather of a job of several (a check of that code on one GPU; not the headline measurement)")
    ap.add_argument("--frames-per-gather", type=int, default=0,
                    help="frames whose strips travel in one all-gather (jobs of several ranks; 0: per workload, 8 at cfg1)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="frames kept in flight for the extra `pipelined` figure (0: skip it); `value` itself is always "
                         "measured one frame after the other on one stream")
    args = ap.parse_args()

    b = Bench(args)
    if args.workload in BANDED:       # on its own (profiling): not a headline line
        if b.world != 1:
            sys.exit("bench.py: the banded workloads run on one GPU")
        r = b.run_banded(args.workload, min(args.steps, OTHER_STEPS[args.workload]), min(args.warmup, 1))
        print(json.dumps(dict(r, metric="camera-ray x beam radiance queries/sec (Mray/s)", n_gpus=1)), flush=True)
        b.close()
        return
    head = b.run(args.workload, args.steps, args.warmup, with_extras=True)
    others = {}
    if not args.only:
        # with several ranks only the configurations whose frames are long enough to be worth sharding
        names = [n for n in sorted(WORKLOADS) if n != args.workload and (b.world == 1 or n in ("cfg2", "cfg4"))]
        for n in names:
            r = b.run(n, min(OTHER_STEPS[n], args.steps), min(5, args.warmup), with_extras=False)
            if r is not None:
                others[n] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                             "config": r["config"], "kernel_ms": r["roofline"]["kernel_ms"],
                             "roofline_frac": r["roofline"]["frac"], "roofline_achieved": r["roofline"]["achieved"],
                             "flop_per_frame": r["roofline"]["flop_per_frame"], "flop_source": r["roofline"]["flop_source"],
                             "per_ray": r["per_ray"], "phases_ms": r["phases_ms"], "frame_ok": r["frame_ok"],
                             "roofline": r["roofline"], "phases_gb_per_s": r["phases_gb_per_s"]}
        if b.world == 1 and not args.no_banded:
            for n in sorted(BANDED):
                others[n] = b.run_banded(n, OTHER_STEPS[n], 1)
    if b.rank == 0:
        line = {"metric": "camera-ray x beam radiance queries/sec (Mray/s) at 512^2 Cornell box, 1-8 GPUs",
                "value": head["value"], "unit": "Mray/s", "n_gpus": b.world, "steps": head["steps"], "warmup": head["warmup"],
                "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        line.update({k: v for k, v in head.items() if k not in line})
        line.setdefault("cpu_baseline", None)
        line["other_workloads"] = others
        print(json.dumps(line), flush=True)
    b.close()


if __name__ == "__main__":
    main()
