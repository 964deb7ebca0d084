"""Host time of one frame call (ctypes + the library's host code + the launches): frames of a tiny image are queued
without waiting, so the wall time per frame is the host's unless the GPU is slower (the GPU time is printed beside it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as pb
for W, H, photons in ((64, 64, 50), (512, 64, 1000), (512, 512, 1000)):
    with pb.Context(pb.control_defaults(screen_width=W, screen_height=H, photon_number=photons)) as ctx:
        ctx.set_scene(); ctx.cast_photon_beams()
        out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            ctx.draw_strips_device(out.data_ptr(), 0, 1, H // 8, cull="bins", stream=s)
        torch.cuda.synchronize()
        n = 3000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter(); e0.record()
        for _ in range(n):
            ctx.draw_strips_device(out.data_ptr(), 0, 1, H // 8, cull="bins", stream=s)
        t_host = (time.perf_counter() - t) / n * 1e6
        e1.record(); torch.cuda.synchronize()
        t_all = (time.perf_counter() - t) / n * 1e6
        print(f"{W}x{H}, {photons} photons: host {t_host:.2f} us per call to queue, {t_all:.2f} us per frame in all, device {e0.elapsed_time(e1) / n * 1e3:.2f} us", flush=True)
