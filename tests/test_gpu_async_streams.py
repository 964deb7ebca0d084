"""Calls of one context queued on different streams and never waited for by the host: upload on the caller's stream,
LBVH build on the context's own, frames on the caller's.  The library orders them on the device (the build waits for
the preparation of the set it reads and for the frames that still traverse the tree it replaces; an upload waits for
the frames that still read the records it overwrites), so the frames must equal those of the same calls with a host
wait after each."""
import numpy as np
import pytest

import raytracer_amd as pb

pytestmark = pytest.mark.gpu

W = H = 256


def _segment_sets(torch, n_sets):
    sets = []
    with pb.Context(pb.control_defaults(screen_width=W, screen_height=H, photon_number=6000, bounces=2), device=0) as src:
        src.set_scene()
        for k in range(n_sets):
            src.cast_photon_beams(seed=100 + k)
            segs = src.download_segments()
            lo, hi = src.root_box()
            dev = torch.from_numpy(segs.view(np.uint8).copy()).cuda()
            sets.append((dev, len(segs), lo, hi, segs))
    return sets


@pytest.mark.parametrize("cull", ["lbvh", "bins"])
def test_upload_build_frame_loop_without_host_waits(cull):
    import torch
    sets = _segment_sets(torch, 3)
    with pb.Context(pb.control_defaults(screen_width=W, screen_height=H), device=0) as ctx:
        ctx.set_scene()
        want = []
        for dev, n, lo, hi, segs in sets:                         # every call waited for
            ctx.upload_segments(segs, lo, hi)
            if cull == "lbvh":
                ctx.build_tree()
            torch.cuda.synchronize()
            want.append(ctx.draw(0, 0, W, H, cull=cull).copy())
        assert not np.array_equal(want[0], want[1])
        rounds = 8
        out = torch.full((rounds, len(sets), H, W, 3), -1.0, dtype=torch.float32, device="cuda")
        user = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(user):
            for r in range(rounds):
                for k, (dev, n, lo, hi, _) in enumerate(sets):
                    ctx.upload_segments_device(dev.data_ptr(), n, lo, hi, stream=user.cuda_stream)
                    if cull == "lbvh":
                        ctx.build_tree()                              # the context's own stream; returns at once
                    ctx.draw_device(out[r, k].data_ptr(), 0, 0, W, H, cull=cull, stream=user.cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for r in range(rounds):
            for k in range(len(sets)):
                assert np.array_equal(got[r, k], want[k]), (r, k)


def test_upload_on_one_stream_while_frames_run_on_another():
    import torch
    sets = _segment_sets(torch, 2)
    with pb.Context(pb.control_defaults(screen_width=W, screen_height=H), device=0) as ctx:
        ctx.set_scene()
        want = []
        for dev, n, lo, hi, segs in sets:
            ctx.upload_segments(segs, lo, hi)
            want.append(ctx.draw(0, 0, W, H, cull="bins").copy())
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        rounds = 6
        out = torch.full((rounds, 2, H, W, 3), -1.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for r in range(rounds):
            for k, (dev, n, lo, hi, _) in enumerate(sets):
                up, fr = a, b                                     # the uploading stream is never the drawing one
                ctx.upload_segments_device(dev.data_ptr(), n, lo, hi, stream=up.cuda_stream)
                for _ in range(3):                                # several frames queued behind the upload
                    ctx.draw_device(out[r, k].data_ptr(), 0, 0, W, H, cull="bins", stream=fr.cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for r in range(rounds):
            for k in range(2):
                assert np.array_equal(got[r, k], want[k]), (r, k)
