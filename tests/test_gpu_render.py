"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle and the
reference's committed render.  Tolerances are those stated for the radiance path:
max-abs <= 2e-3 * max(1, |ref|) and RMSE <= 1e-4 per channel on pre-quantisation radiance,
<= 1 LSB on >= 99.9 % of pixels on the 8-bit image."""
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pb():
    import raytracer_amd as pb
    if pb.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path cannot run (there is no CPU fallback)")
    return pb


@pytest.fixture(scope="module")
def cfg1():
    o = orc.Oracle()
    o.cast_photon_beams(); o.bound_photon_beams(); o.build_tree()
    return o


def check_radiance(gpu, ref, what):
    gpu = gpu.astype(np.float64)
    tol = 2e-3 * np.maximum(1.0, np.abs(ref))
    err = np.abs(gpu - ref)
    assert (err <= tol).all(), f"{what}: max-abs error {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    rmse = np.sqrt((err ** 2).mean(axis=(0, 1)))
    assert (rmse <= 1e-4).all(), f"{what}: RMSE per channel {rmse}"
    return err.max(), rmse.max()


def test_config1_tree_segments_full_frame(pb, cfg1, golden_dir):
    """Configuration 1 (512^2, 1000 photons, SEED 42): the 987 tree segments with their leaf-node
    gate boxes; GPU radiance vs oracle, and the quantised image vs the reference's screenshot."""
    from PIL import Image
    segs = cfg1.export_segments(tree=True)
    lo, hi = cfg1.root_box()
    tri, _ = orc.load_test_model()
    ref, _ = orc.render_segments(cfg1.control, orc.tri_xyz(tri), orc.transformation_matrix(), segs, lo, hi,
                                 0, 0, 512, 512)
    with pb.Context(pb.control_defaults()) as ctx:
        ctx.set_scene()
        ctx.upload_segments(segs, lo, hi)
        img = ctx.draw()                              # default arithmetic (fast)
        ctx.enable_counters(True)
        ctx.draw()
        cnt = ctx.counters()
        ctx.set_precise(True)                         # the reference's exact operation order
        img_precise = ctx.draw()
        cnt_precise = ctx.counters()
        ctx.enable_counters(False)
        ctx.set_precise(False)
        ctx.build_tree()
        img_lbvh = ctx.draw(use_lbvh=True)
        # screen-space bins + fixed-point sums: same pairs, order-independent, reproducible bit for bit
        img_bins = ctx.draw(cull="bins")
        ctx.enable_counters(True)
        img_bins2 = ctx.draw(cull="bins")
        cnt_bins = ctx.counters()
        ctx.set_precise(True)
        img_bins_precise = ctx.draw(cull="bins")
        cnt_bins_precise = ctx.counters()
        ctx.set_precise(False)
        ctx.enable_counters(False)
    maxerr, rmse = check_radiance(img, ref, "config 1")
    check_radiance(img_lbvh, ref, "config 1 through the LBVH")
    berr, brmse = check_radiance(img_bins, ref, "config 1 through screen bins")
    bperr, _ = check_radiance(img_bins_precise, ref, "config 1 through screen bins, exact operation order")
    print(f"config 1 bins: max-abs {berr:.3e} rmse {brmse:.3e}; exact order {bperr:.3e}; counters {cnt_bins}")
    assert np.array_equal(img_bins, img_bins2)
    assert bperr < 1e-4
    assert cnt_bins_precise["valid_hits"] == 3144103 and cnt_bins["shortcut_wrong"] == 0
    assert cnt_bins["valid_hits"] == cnt["valid_hits"]
    perr, prmse = check_radiance(img_precise, ref, "config 1, exact operation order")
    print(f"config 1: fast max-abs {maxerr:.3e} rmse {rmse:.3e}; exact-order max-abs {perr:.3e} rmse {prmse:.3e}")
    print("counters", cnt)
    assert perr < 1e-4 and prmse < 5e-6            # only libm's expf differs from the host
    assert cnt["rays"] == 262144 and cnt["rays_hit_scene"] == 262052
    assert cnt_precise["valid_hits"] == 3144103    # exactly the reference traversal's valid beam hits
    # grazing rays and the ~1e-7-long tail segments may fall either side of disc = 0 or t0 > 0
    assert abs(cnt["valid_hits"] - 3144103) <= 3144103 * 5e-4
    assert cnt["shortcut_wrong"] == 0              # the gate shortcut never contradicts HitBoundingBox
    assert cnt["gate_passes"] == 47768919          # the reference's HitCylinder call count (N_cull)
    argb = pb.quantise_argb(img)
    shot = np.array(Image.open(os.path.join(golden_dir, "reference_screenshot.png")).convert("RGBA")).astype(np.int32)
    mine = np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], -1).astype(np.int32)
    d = np.abs(mine - shot[..., :3]).max(-1)
    assert d.max() <= 1 and (d > 0).mean() <= 1e-3, (d.max(), int((d > 0).sum()))


def test_scene_setup_matches_oracle(pb):
    tri, nrm = pb.load_test_model()
    otri, onrm = orc.load_test_model()
    assert np.array_equal(tri, orc.tri_xyz(otri))
    assert np.array_equal(nrm, onrm[:, :3])
    assert np.array_equal(pb.transformation_matrix(), orc.transformation_matrix())


def test_ragged_rectangles_and_strips(pb, cfg1):
    """Rectangles that are not multiples of the 8x8 tile, a 1x1 rectangle, and interleaved strips."""
    import torch
    segs = cfg1.export_segments(tree=True)
    lo, hi = cfg1.root_box()
    with pb.Context(pb.control_defaults()) as ctx:
        ctx.set_scene()
        ctx.upload_segments(segs, lo, hi)
        full = ctx.draw()
        full_bins = ctx.draw(cull="bins")
        for (x0, y0, w, h) in [(250, 90, 13, 5), (0, 0, 1, 1), (505, 500, 7, 12), (256, 93, 9, 1)]:
            part = ctx.draw(x0, y0, w, h)
            assert np.array_equal(part, full[y0:y0 + h, x0:x0 + w]), (x0, y0, w, h)
            part = ctx.draw(x0, y0, w, h, cull="bins")
            assert np.array_equal(part, full_bins[y0:y0 + h, x0:x0 + w]), ("bins", x0, y0, w, h)
        out = torch.zeros((8 * 16, 512, 3), dtype=torch.float32, device="cuda")
        ctx.draw_strips_device(out.data_ptr(), 3, 4, 16, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for k in range(16):
            y = 8 * (3 + 4 * k)
            assert np.array_equal(got[8 * k:8 * k + 8], full[y:y + 8]), k
        ctx.draw_strips_device(out.data_ptr(), 1, 8, 8, stream=torch.cuda.current_stream().cuda_stream, cull="bins")
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for k in range(8):
            y = 8 * (1 + 8 * k)
            assert np.array_equal(got[8 * k:8 * k + 8], full_bins[y:y + 8]), ("bins", k)


def test_empty_segment_set_and_errors(pb):
    with pb.Context(pb.control_defaults(screen_width=64, screen_height=64)) as ctx:
        with pytest.raises(pb.PhotonBeamsError):
            ctx.draw()                               # no scene yet
        ctx.set_scene()
        with pytest.raises(pb.PhotonBeamsError):
            ctx.draw()                               # no segments yet
        ctx.upload_segments(np.zeros(0, pb.SEGMENT_DTYPE), [-1, -1, 1], [1, 1, 4])
        img = ctx.draw()
        assert img.shape == (64, 64, 3) and not img.any()
        img = ctx.draw(cull="bins")
        assert img.shape == (64, 64, 3) and not img.any()
        with pytest.raises(pb.PhotonBeamsError):
            ctx.draw(0, 0, 65, 1)                    # rectangle exceeds the screen


def test_degenerate_segments_are_harmless(pb):
    """Records a caller can upload but the photon pass never makes: zero length, exactly along +y / -y
    (findRotationMatrix, Intersections.h:499-513, is singular for -y), behind the camera, zero and huge radius,
    NaN and infinite coordinates, an inverted gate box.  Mixed into a normal segment set they must neither
    disturb the other segments' radiance nor depend on how candidates are found."""
    kw = dict(photon_number=300, bounces=1, screen_width=64, screen_height=64)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        good = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        ctx.set_precise(True)
        clean = ctx.draw(cull="scan").astype(np.float64)
        bad = np.zeros(10, pb.SEGMENT_DTYPE)
        bad["radius"] = 0.03; bad["energy"] = 0.5; bad["seg_id"] = 1
        bad["cull_lo"] = [-1, -1, 2]; bad["cull_hi"] = [1, 1, 4]
        P = np.array([0.1, -0.2, 3.0], np.float32)
        bad["start"] = P; bad["end"] = P + [0.3, 0.4, 0.1]; bad["orig_start"] = P
        bad["end"][0] = P                                              # zero length
        bad["end"][1] = P + [0, -0.7, 0]                               # exactly -y
        bad["end"][2] = P + [0, 0.7, 0]                                # exactly +y
        bad["start"][3] = [0.2, 0.1, -2.0]; bad["end"][3] = [0.2, 0.3, -1.0]   # behind the camera
        bad["radius"][4] = 0.0
        bad["radius"][5] = 10.0
        bad["start"][6] = [np.nan, 0, 3]
        bad["end"][7] = [np.inf, 0, 3]
        bad["cull_lo"][8] = [1, 1, 4]; bad["cull_hi"][8] = [-1, -1, 2]  # inverted gate box: never entered
        bad["energy"][9] = [np.nan, np.nan, np.nan]; bad["cull_lo"][9] = [5, 5, 5]; bad["cull_hi"][9] = [6, 6, 6]
        inert = [0, 1, 3, 4, 6, 7, 8, 9]       # cannot contribute: no cylinder, NaN frame, or a gate that is never passed
        mixed = np.concatenate([good[:100], bad[inert], good[100:]])
        ctx.upload_segments(mixed, lo, hi)
        ctx.build_tree()
        frames = {c: ctx.draw(cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
        # every record, including the huge and the +y one: the searches must still agree with each other
        every = np.concatenate([good, bad])
        ctx.upload_segments(every, lo, hi)
        ctx.build_tree()
        all3 = {c: ctx.draw(cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
    for name, img in frames.items():
        assert np.isfinite(img).all(), name
        np.testing.assert_allclose(img, clean, rtol=1e-6, atol=1e-9, err_msg=name)
    fin = np.isfinite(all3["scan"])
    assert fin.mean() > 0.99
    np.testing.assert_allclose(all3["lbvh"][fin], all3["scan"][fin], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(all3["bins"][fin], all3["scan"][fin], rtol=1e-5, atol=1e-7)
    tri, _ = orc.load_test_model()
    ref, _ = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(),
                                 every, lo, hi, 0, 0, 64, 64)
    both = fin & np.isfinite(ref)
    err = np.abs(all3["scan"] - ref)[both]
    assert both.mean() > 0.99 and (err <= 2e-5 * np.maximum(1.0, np.abs(ref[both]))).all(), err.max()


def test_overweight_and_nan_weight_segments_saturate_instead_of_corrupting_tiles(pb):
    """The binned path sums contributions as fixed-point integers, valid below 4096 each.  An uploaded beam whose
    weight energy / r^2 is far beyond that must saturate near PB_FIX_LIMIT = 2047 on the pixels it covers and
    leave every other pixel of its tiles alone; a NaN energy weighs nothing.  (The reference would show an
    over-bright or NaN pixel there: both are outside the display range.)"""
    kw = dict(photon_number=300, bounces=1, screen_width=64, screen_height=64)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        good = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        clean = ctx.draw(cull="bins").astype(np.float64)
        bad = np.zeros(2, pb.SEGMENT_DTYPE)
        bad["radius"] = 0.02; bad["seg_id"] = 1
        bad["cull_lo"] = [-1, -1, 2]; bad["cull_hi"] = [1, 1, 4]
        bad["start"] = [-0.4, -0.3, 2.6]; bad["end"] = [0.3, 0.2, 2.9]; bad["orig_start"] = bad["start"]
        bad["energy"][0] = [1e12, 1e12, 2e12]
        bad["start"][1] = [-0.4, 0.3, 2.6]; bad["end"][1] = [0.3, -0.2, 2.9]; bad["orig_start"][1] = bad["start"][1]
        bad["energy"][1] = [np.nan, np.nan, np.nan]
        ctx.upload_segments(np.concatenate([good, bad[1:]]), lo, hi)
        ctx.build_tree()
        nan_weight = ctx.draw(cull="bins").astype(np.float64)
        ctx.upload_segments(np.concatenate([good[:50], bad, good[50:]]), lo, hi)
        ctx.build_tree()
        bins = ctx.draw(cull="bins").astype(np.float64)
        scan = ctx.draw(cull="scan").astype(np.float64)     # floating-point sums: no saturation
    np.testing.assert_array_equal(nan_weight, clean)
    assert np.isfinite(bins).all()
    over = scan.max(axis=2) > 2000.0
    assert 10 < over.sum() < 0.2 * over.size
    assert (bins[over].max(axis=1) >= 2000.0).all() and (bins[over] < 3 * 2048.0).all()
    np.testing.assert_allclose(bins[~over], scan[~over], rtol=1e-5, atol=1e-7)


def test_large_frame_takes_the_multi_kernel_queue_path(pb):
    """Frames above 16384 tiles build the evaluation queue with the multi-kernel path (device-wide
    scans) instead of the single-block one: 1160 x 1096 pixels = 145 x 137 tiles.  All three candidate
    searches must agree, and a sparse lattice of pixels must match the oracle."""
    kw = dict(photon_number=300, bounces=1, screen_width=1160, screen_height=1096)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        scan = ctx.draw(cull="scan").astype(np.float64)
        lbvh = ctx.draw(cull="lbvh").astype(np.float64)
        bins = ctx.draw(cull="bins").astype(np.float64)
        bins2 = ctx.draw(cull="bins")
        # a small rectangle next (single-block queue, counters handed over between frames), then the large
        # frame again: the two paths share the workspace
        small = [ctx.draw(400, 300, 256, 200, cull="bins") for _ in range(2)]
        bins3 = ctx.draw(cull="bins")
    assert np.array_equal(small[0], bins2[300:500, 400:656]) and np.array_equal(small[1], small[0])
    assert np.array_equal(bins3, bins2)
    np.testing.assert_allclose(lbvh, scan, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(bins, scan, rtol=1e-5, atol=1e-7)
    assert np.array_equal(bins.astype(np.float32), bins2)
    tri, _ = orc.load_test_model()
    ref, cnt = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(),
                                   segs, lo, hi, 3, 5, 72, 68, step_x=16, step_y=16)
    got = bins[5::16, 3::16][:68, :72]
    err = np.abs(got - ref)
    assert (err <= 2e-3 * np.maximum(1.0, np.abs(ref))).all(), err.max()
    assert np.sqrt((err ** 2).mean()) <= 1e-4 and ref.max() > 0.1


@pytest.mark.parametrize("ssaa", [2, 3])
def test_supersampling_matches_oracle(pb, ssaa):
    """SSAA > 1 (skeleton.cpp:336-361): SSAA^2 sub-rays per pixel, focal = SCREEN_WIDTH * SSAA, the sum
    divided by SSAA (not SSAA^2, as the reference does)."""
    kw = dict(photon_number=300, bounces=1, screen_width=96, screen_height=80, ssaa=ssaa)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        frames = {c: ctx.draw(cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
        part = ctx.draw(8, 16, 40, 24, cull="bins").astype(np.float64)
    tri, _ = orc.load_test_model()
    ref, cnt = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(),
                                   segs, lo, hi, 0, 0, 96, 80)
    assert cnt["rays"] == 96 * 80 * ssaa * ssaa and ref.max() > 0.1
    for name, img in frames.items():
        err = np.abs(img - ref)
        assert (err <= 2e-3 * np.maximum(1.0, np.abs(ref))).all(), (name, err.max())
        assert np.sqrt((err ** 2).mean()) <= 1e-4, name
    np.testing.assert_allclose(part, frames["bins"][16:40, 8:48], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("world", [2, 8])
def test_rank_strips_reassemble_to_the_single_gpu_frame(pb, world):
    """The N-GPU path of bench.py in one process: every rank's strips (tiles.strips_of_rank) rendered by
    draw_strips_device with the segment set uploaded from a device buffer, then put in image order by
    tiles.assemble - bit-identical to the frame one GPU renders (binned mode, fixed-point sums)."""
    import torch
    from raytracer_amd import tiles
    kw = dict(photon_number=1500, bounces=2, screen_width=256, screen_height=256)
    H = W = 256
    with pb.Context(pb.control_defaults(**kw)) as src:
        src.set_scene()
        src.cast_photon_beams()
        segs = src.download_segments()
        lo, hi = src.root_box()
        full = src.draw(cull="bins")
    stream = torch.cuda.current_stream().cuda_stream
    d_segs = torch.from_numpy(segs.view(np.uint8).reshape(-1).copy()).cuda()
    first, stride, n_my = tiles.strips_of_rank(H, 0, world)
    gathered = torch.zeros((world, n_my * 8, W, 3), dtype=torch.float32, device="cuda")
    for rank in range(world):
        first, stride, n_my = tiles.strips_of_rank(H, rank, world)
        with pb.Context(pb.control_defaults(**kw)) as ctx:
            ctx.set_scene()
            ctx.upload_segments_device(d_segs.data_ptr(), len(segs), lo, hi, stream=stream)
            torch.cuda.synchronize()
            ctx.draw_strips_device(gathered[rank].data_ptr(), first, stride, n_my, cull="bins", stream=stream)
            torch.cuda.synchronize()
    frame = tiles.assemble(gathered, torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"))
    assert np.array_equal(frame.cpu().numpy(), full)
    assert full.max() > 1.0


def test_one_dimensional_homogeneous_estimator(pb):
    """ONE_DIMENSIONAL in a homogeneous medium: BeamRadiance's Integral_73 branch (BeamRadiance.h:510-524).
    Double arithmetic in the reference's order: the frame agrees to the float32 rounding of the output."""
    kw = dict(photon_number=800, bounces=1, screen_width=128, screen_height=128, one_dimensional=1)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        frames = {c: ctx.draw(mode=pb.MODE_73, cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
        std = ctx.draw(mode=pb.MODE_721, cull="bins")
    tri, _ = orc.load_test_model()
    ref, cnt = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(),
                                   segs, lo, hi, 0, 0, 128, 128, mode=orc.MODE_73)
    assert ref.max() > 0.05 and cnt["valid_hits"] > 50000
    for name, img in frames.items():
        err = np.abs(img - ref)
        assert (err <= 1e-6 * np.maximum(1.0, np.abs(ref))).all(), (name, err.max())
    assert not np.allclose(std, frames["bins"], rtol=0.05)     # a different estimator, not the 2D one


def test_adaptive_cones_generic_estimator(pb):
    """ADAPTIVE: first-cast beams widen into cones, BeamRadiance (BeamRadiance.h:471-555) sends them through
    HitCone + Integral_721_ada and scattered beams through HitCylinder + Integral_721.  The Riemann sum is in
    double on both sides; 1/r^2 near a cone's apex makes single pixels huge, so the comparison is on the
    displayed range (PutPixelSDL clamps at 1) and, relative, below the fixed-point saturation."""
    kw = dict(photon_number=300, bounces=1, screen_width=96, screen_height=96, adaptive=1)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        frames = {c: ctx.draw(mode=pb.MODE_DEFAULT, cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
        cyl = ctx.draw(mode=pb.MODE_721, cull="bins")
    assert (segs["flags"] & 1).sum() > 200 and (segs["flags"] & 1).sum() < len(segs)
    tri, _ = orc.load_test_model()
    ref, cnt = orc.render_segments(orc.control_defaults(**kw), orc.tri_xyz(tri), orc.transformation_matrix(),
                                   segs, lo, hi, 0, 0, 96, 96, mode=orc.MODE_GENERIC)
    assert cnt["valid_hits"] > 20000 and np.isfinite(ref).all() and (ref > 0).mean() > 0.3
    for name, img in frames.items():
        small = ref < 1000.0
        err = np.abs(img - ref)
        # the scattered beams' Integral_721 runs in float: device and host expf differ in the last bit
        assert (err[small] <= 2e-5 + 1e-4 * np.abs(ref[small])).all(), (name, err[small].max())
        assert (err[small] <= 1e-6 * np.maximum(1.0, np.abs(ref[small]))).mean() > 0.98, name
        a, b = pb.quantise_argb(img), pb.quantise_argb(ref)
        assert (a != b).mean() < 1e-3, name
    assert not np.allclose(cyl, frames["bins"], rtol=0.05)


def test_short_view_rays(pb):
    """SHORT_VIEW (skeleton.cpp:442-449): each camera ray ends at a free-flight distance drawn for it, which
    moves the limit of every gate.  The device keys the draw by the sub-pixel; the oracle makes the same draws.
    Strong extinction so that most rays end in the medium."""
    kw = dict(photon_number=600, bounces=1, screen_width=128, screen_height=96, short_view=1, ssaa=2,
              absorption_c=0.3, scattering_c=0.1)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        ctx.build_tree()
        ctx.set_precise(True)
        frames = {c: ctx.draw(mode=pb.MODE_721, cull=c).astype(np.float64) for c in ("scan", "lbvh", "bins")}
        ctx.set_precise(False)
        fast = ctx.draw(mode=pb.MODE_721, cull="bins").astype(np.float64)
    tri, _ = orc.load_test_model()
    args = (orc.tri_xyz(tri), orc.transformation_matrix(), segs, lo, hi, 0, 0, 128, 96)
    ref, cnt = orc.render_segments(orc.control_defaults(**kw), *args, mode=orc.MODE_721)
    full, cnt_full = orc.render_segments(orc.control_defaults(**dict(kw, short_view=0)), *args, mode=orc.MODE_721)
    assert cnt["valid_hits"] > 1000 and cnt["valid_hits"] < 0.8 * cnt_full["valid_hits"]
    for name, img in frames.items():
        assert np.abs(img - ref).max() <= 2e-5 + 1e-4 * ref.max(), name
    assert np.abs(fast - ref).max() <= 1e-4 * max(1.0, ref.max())


@pytest.mark.parametrize("kw", [dict(screen_width=512, screen_height=512, photon_number=1000, bounces=0),
                                dict(screen_width=200, screen_height=136, photon_number=600, bounces=2, ssaa=2)])
def test_repeated_frames_of_one_segment_set_are_bit_identical(pb, kw):
    """The first binned frame of a segment set sizes the workspace and leaves the lists, their offsets and the item
    queue in it; the frames after it are two launches (ray set-up, evaluation whose last item per tile writes the
    pixels).  Every such frame must equal the first one and the ordered scan path's radiance."""
    import torch
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        W, H = kw["screen_width"], kw["screen_height"]
        first = ctx.draw(cull="bins")
        assert first.max() > 0
        out = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for k in range(30):
            ctx.draw_device(out.data_ptr(), 0, 0, W, H, cull="bins", stream=stream)
            if k % 10 == 9:
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), first), k
                out.fill_(-1.0)
        scan = ctx.draw(cull="scan")
        assert np.abs(scan - first).max() <= 1e-6 * max(1.0, float(scan.max()))
        # a part of the frame, then the whole again: each change of geometry sizes anew
        part = ctx.draw(40, 24, 99, 77, cull="bins")
        assert np.array_equal(part, first[24:24 + 77, 40:40 + 99])
        assert np.array_equal(ctx.draw(cull="bins"), first)
        assert np.array_equal(ctx.draw(cull="bins"), first)
        # nothing kept from frame to frame (counting pass and queue build every frame, what bench.py reports as
        # schedule_rebuilt_every_frame), then kept again
        import os
        os.environ["PB_TUNE_NO_KEEP"] = "1"
        pb.reload_tuning()
        try:
            for _ in range(3):
                ctx.draw_device(out.data_ptr(), 0, 0, W, H, cull="bins", stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), first)
        finally:
            del os.environ["PB_TUNE_NO_KEEP"]
            pb.reload_tuning()
        out.fill_(-1.0)
        ctx.draw_device(out.data_ptr(), 0, 0, W, H, cull="bins", stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), first)
        # the lists filled again every frame beside the ray set-up (PB_TUNE_REFILL), then kept again
        os.environ["PB_TUNE_REFILL"] = "1"
        pb.reload_tuning()
        try:
            for _ in range(2):
                out.fill_(-1.0)
                ctx.draw_device(out.data_ptr(), 0, 0, W, H, cull="bins", stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), first)
        finally:
            del os.environ["PB_TUNE_REFILL"]
            pb.reload_tuning()
        # a new segment set in the same context and geometry: nothing of the old lists may survive
        ctx.cast_photon_beams(seed=7)
        other = ctx.draw(cull="bins")
        assert not np.array_equal(other, first)
        again = ctx.draw(cull="bins")                 # lists kept from the frame before
        assert np.array_equal(again, other)
        scan2 = ctx.draw(cull="scan")
        assert np.abs(scan2 - other).max() <= 1e-6 * max(1.0, float(scan2.max()))
        assert np.array_equal(ctx.draw(cull="bins"), other)     # after a frame of another cull mode on the same workspace


@pytest.mark.parametrize("het", [False, True])
def test_both_binning_passes_give_the_same_frame(pb, het, monkeypatch):
    """Segment sets above a limit (2^20) are binned by the walk over each segment's own rectangle of macro tiles and
    the multi-kernel queue, smaller ones by the lane-per-macro-tile pass: lowered to 0, the limit sends a small set
    down the first way.  The sums are fixed point, so the frames must agree bit for bit."""
    import os
    kw = dict(photon_number=1500, bounces=2, screen_width=256, screen_height=256)
    if het:
        kw.update(heterogeneous=1, one_dimensional=1)
    mode = pb.MODE_73HET if het else pb.MODE_721
    frames = {}
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        if het:
            ctx.set_grid(*pb.make_grid())
            ctx.set_precise(True)          # the exact-order estimator runs behind either pass
        ctx.cast_photon_beams()
        for limit in ("", "0", ""):
            if limit:
                monkeypatch.setenv("PB_TUNE_FUSED_MAX", limit)
            else:
                monkeypatch.delenv("PB_TUNE_FUSED_MAX", raising=False)
            assert os.environ.get("PB_TUNE_FUSED_MAX", "") == limit
            pb.reload_tuning()             # the switches are read when a context is created, not by its frames
            frames.setdefault(limit, []).extend(ctx.draw(mode=mode, cull="bins") for _ in range(2))
    ref = frames[""][0]
    assert ref.max() > 0.05 and np.isfinite(ref).all()
    for f in frames[""][1:] + frames["0"]:
        assert np.array_equal(f, ref)


@pytest.mark.parametrize("ssaa", [1, 2])
def test_kept_queue_of_survivors_gives_the_same_frames(pb, ssaa, monkeypatch):
    """The frame after the one that sizes the workspace tests every item's candidates against its tile once, keeps the
    survivors and cuts the queue anew by survivors; later frames start at the pair loop.  Whatever the item length,
    with or without that step, for either arithmetic and for ragged rectangles the frame is the same, bit for bit
    (fixed-point sums do not depend on how a tile's pairs are dealt to items)."""
    import torch
    W, H = 200, 136
    kw = dict(screen_width=W, screen_height=H, ssaa=ssaa, photon_number=4000, bounces=2)
    stream = torch.cuda.current_stream().cuda_stream

    def frames(ctx, n, rect=(0, 0, W, H)):
        x0, y0, w, h = rect
        out = torch.full((n, h, w, 3), -1.0, dtype=torch.float32, device="cuda")
        for i in range(n):
            ctx.draw_device(out[i].data_ptr(), x0, y0, w, h, cull="bins", stream=stream)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    monkeypatch.setenv("PB_TUNE_NO_PREFINE", "1")
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene(); ctx.cast_photon_beams()
        plain = frames(ctx, 2)
        ctx.set_precise(True)
        plain_precise = frames(ctx, 2)
        part_plain = frames(ctx, 2, (24, 16, 99, 77))
    monkeypatch.delenv("PB_TUNE_NO_PREFINE")
    assert plain.max() > 0 and np.array_equal(plain[0], plain[1])
    for length in (None, "5", "64", "256", "one launch", "two launches"):
        # frames of few items per tile are one launch whose items set up their own rays; forced on and off here
        monkeypatch.delenv("PB_TUNE_INLINE", raising=False)
        if length == "one launch":
            monkeypatch.setenv("PB_TUNE_INLINE", "1000"); monkeypatch.setenv("PB_TUNE_SURV", "24")
        elif length == "two launches":
            monkeypatch.setenv("PB_TUNE_INLINE", "0"); monkeypatch.setenv("PB_TUNE_SURV", "24")
        elif length is None:
            monkeypatch.delenv("PB_TUNE_SURV", raising=False)
        else:
            monkeypatch.setenv("PB_TUNE_SURV", length)
        with pb.Context(pb.control_defaults(**kw)) as ctx:
            ctx.set_scene(); ctx.cast_photon_beams()
            got = frames(ctx, 4)          # sizing frame, the frame that makes the lists, two that use them
            for i in range(4):
                assert np.array_equal(got[i], plain[0]), (length, i)
            ctx.set_precise(True)         # the exact-order kernels walk the same kept queue
            for f in frames(ctx, 2):
                assert np.array_equal(f, plain_precise[0]), length
            for f in frames(ctx, 3, (24, 16, 99, 77)):
                assert np.array_equal(f, part_plain[0]), length
            ctx.set_precise(False)
            ctx.enable_counters(True)     # a counted frame builds its queue anew; the next frames keep one again
            counted = frames(ctx, 1)
            ctx.enable_counters(False)
            np.testing.assert_allclose(counted[0], plain[0], rtol=1e-5, atol=1e-7)
            for f in frames(ctx, 3):
                assert np.array_equal(f, plain[0]), length
