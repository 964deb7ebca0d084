"""pb_render_noalloc_d (the entry point that allocates nothing: its workspace is a file-scope device array)
against pb_render_d through the C ABI: bit-identical frames on two calls in a row, the second on a smaller
input; sizes beyond the stated maxima are refused with a message."""
import numpy as np
import pytest

import raytracer_amd as pb

pytestmark = pytest.mark.gpu


def _both(ctx, torch, x0, y0, w, h, mode=pb.MODE_721):
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    b = torch.full((h, w, 3), -2.0, dtype=torch.float32, device="cuda")
    ctx.draw_device(a.data_ptr(), x0, y0, w, h, mode=mode, cull="bins", stream=stream)
    ctx.draw_noalloc_device(b.data_ptr(), x0, y0, w, h, mode=mode, stream=stream)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_frames_equal_pb_render_d_twice_in_a_row():
    import torch
    kw = dict(screen_width=512, screen_height=512, photon_number=3000, bounces=2)
    with pb.Context(pb.control_defaults(**kw), device=0) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        a, b = _both(ctx, torch, 0, 0, 512, 512)
        assert a.max() > 0 and np.isfinite(a).all()
        assert np.array_equal(a, b), "first call: frames differ"
        # the same frame again (the verified path: no read-back), then a smaller input: a third of the segments,
        # a ragged rectangle
        a1, b1 = _both(ctx, torch, 0, 0, 512, 512)
        assert np.array_equal(a1, a) and np.array_equal(b1, a)
        ctx.upload_segments(segs[: len(segs) // 3], lo, hi)
        a2, b2 = _both(ctx, torch, 40, 24, 203, 117)
        assert a2.max() > 0 and np.array_equal(a2, b2), "second call (smaller input): frames differ"
        # and back up to the full set on the same static workspace
        ctx.upload_segments(segs, lo, hi)
        a3, b3 = _both(ctx, torch, 0, 0, 512, 512)
        assert np.array_equal(a3, a) and np.array_equal(b3, a)


def test_noalloc_is_shared_between_contexts_and_estimators():
    import torch
    kw = dict(screen_width=128, screen_height=96, photon_number=400, bounces=1, heterogeneous=1, one_dimensional=1)
    sigma, side = pb.make_grid()
    with pb.Context(pb.control_defaults(**kw), device=0) as het, \
            pb.Context(pb.control_defaults(screen_width=256, screen_height=256, photon_number=500), device=0) as hom:
        het.set_scene(); het.set_grid(sigma, side); het.cast_photon_beams()
        # the static workspace has no room for per-ray walk tables: in a heterogeneous medium the entry point always
        # runs the exact-order estimator, which pb_render_d runs when asked for the reference's operation order
        het.set_precise(True)
        hom.set_scene(); hom.cast_photon_beams()
        for _ in range(2):       # alternate: each call finds the other context's leftovers in the array
            a, b = _both(het, torch, 0, 0, 128, 96, mode=pb.MODE_73HET)
            assert a.max() > 0 and np.array_equal(a, b)
            a, b = _both(hom, torch, 0, 0, 256, 256)
            assert a.max() > 0 and np.array_equal(a, b)


def test_noalloc_contexts_of_one_geometry_do_not_inherit_each_others_lists():
    """Two contexts with the same screen, rectangle and mode, whose segment sets carry the same version number, take
    turns on the shared workspace: what one left there (lists, offsets, item queue) must never serve the other."""
    import torch
    kw = dict(screen_width=256, screen_height=256, photon_number=800, bounces=1)
    with pb.Context(pb.control_defaults(**kw), device=0) as one, pb.Context(pb.control_defaults(**kw), device=0) as two:
        one.set_scene(); one.cast_photon_beams(seed=1)
        two.set_scene(); two.cast_photon_beams(seed=2)
        ref = {}
        for name, ctx in (("one", one), ("two", two)):
            a, b = _both(ctx, torch, 0, 0, 256, 256)
            assert a.max() > 0 and np.array_equal(a, b)
            ref[name] = a
        assert not np.array_equal(ref["one"], ref["two"])
        stream = torch.cuda.current_stream().cuda_stream
        out = torch.zeros((256, 256, 3), dtype=torch.float32, device="cuda")
        for name, ctx in (("one", one), ("one", one), ("two", two), ("two", two), ("one", one), ("two", two)):
            out.fill_(-1.0)
            ctx.draw_noalloc_device(out.data_ptr(), 0, 0, 256, 256, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), ref[name]), name


def test_noalloc_one_dimensional_estimator_equals_pb_render_d():
    """PB_MODE_73 (Integral_73, BeamRadiance.h:112-209): the static workspace has no queues for the pairs the relaxed
    estimator leaves to the exact-order code, so the items evaluate them themselves; the frame is pb_render_d's."""
    import torch
    kw = dict(screen_width=256, screen_height=192, photon_number=1500, bounces=1, one_dimensional=1)
    with pb.Context(pb.control_defaults(**kw), device=0) as ctx:
        ctx.set_scene(); ctx.cast_photon_beams()
        for rect in ((0, 0, 256, 192), (16, 8, 101, 77)):
            a, b = _both(ctx, torch, *rect, mode=pb.MODE_73)
            assert a.max() > 0 and np.isfinite(a).all()
            assert np.array_equal(a, b), rect


def test_noalloc_frame_refused_after_sizing_does_not_poison_the_next():
    """A frame whose lists outgrow the static workspace is refused only after its counting pass has run there.  The
    frame rendered before it (same context, same segment set, a window that fits) must then be sized anew, not taken
    for verified with the refused frame's leftovers."""
    import torch
    kw = dict(screen_width=512, screen_height=512, photon_number=200000, bounces=0)
    with pb.Context(pb.control_defaults(**kw), device=0) as ctx:
        ctx.set_scene(); ctx.cast_photon_beams()
        small = (192, 192, 128, 128)
        a, b = _both(ctx, torch, *small)
        assert a.max() > 0 and np.array_equal(a, b)
        out = torch.zeros((512, 512, 3), dtype=torch.float32, device="cuda")
        with pytest.raises(pb.PhotonBeamsError) as e:
            ctx.draw_noalloc_device(out.data_ptr(), 0, 0, 512, 512)
        assert "(-5)" in str(e.value) and ("list entries" in str(e.value) or "evaluation items" in str(e.value)), str(e.value)
        torch.cuda.synchronize()
        a2, b2 = _both(ctx, torch, *small)
        assert np.array_equal(a2, a) and np.array_equal(b2, a)


def test_noalloc_refuses_frames_beyond_its_maxima():
    import torch
    with pb.Context(pb.control_defaults(screen_width=1024, screen_height=1024, photon_number=200), device=0) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        out = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
        with pytest.raises(pb.PhotonBeamsError) as e:
            ctx.draw_noalloc_device(out.data_ptr(), 0, 0, 1024, 1024)
        assert "(-5)" in str(e.value) and "PB_NOALLOC_MAX_GROUPS" in str(e.value)
        # a 512 x 512 window of the same screen fits
        ctx.draw_noalloc_device(out.data_ptr(), 256, 256, 512, 512)
        torch.cuda.synchronize()
        assert float(out.max().item()) > 0
