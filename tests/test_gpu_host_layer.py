"""The C++ host layer with the reference's function names (raytracer_amd/host) driving StandardRun
(examples/skeleton_headless.cpp) must produce the very image the C ABI gives when driven directly."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "skeleton_headless")


def read_bmp32(path):
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM"
    off = struct.unpack_from("<I", raw, 10)[0]
    w, h = struct.unpack_from("<ii", raw, 18)
    bpp = struct.unpack_from("<H", raw, 28)[0]
    assert bpp == 32 and h > 0
    px = np.frombuffer(raw, dtype="<u4", count=w * h, offset=off).reshape(h, w)
    return px[::-1].copy()        # bottom-up file -> top-down image


def test_standard_run_through_reference_named_functions(tmp_path):
    import raytracer_amd as pb
    if pb.device_count() < 1:
        pytest.fail("no GPU visible")
    assert os.path.exists(EXE), "examples/skeleton_headless is not built (make -C raytracer_amd/csrc)"
    out = str(tmp_path / "shot.bmp")
    r = subprocess.run([EXE, "2000", "2", "256", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout
    assert "Beams size:" in r.stdout and "Render time:" in r.stdout
    img = read_bmp32(out)
    assert img.shape == (256, 256) and (img >> 24 == 128).all()
    with pb.Context(pb.control_defaults(photon_number=2000, bounces=2, screen_width=256, screen_height=256)) as ctx:
        ctx.set_scene()
        n_beams, n_segs = ctx.cast_photon_beams()
        ref = pb.quantise_argb(ctx.draw(cull="bins"))
    assert f"Beams size: {n_beams}" in r.stdout and f"Segment size: {n_segs}" in r.stdout
    assert np.array_equal(img, ref)
    assert (img & 0xFF).max() > 100


def test_bmp_header_equals_the_reference_screenshot(tmp_path):
    """SDL_SaveImage (SDLauxiliary.h:24-53): 122-byte V4 header, BI_BITFIELDS ARGB masks, bottom-up - the
    first 122 bytes of the reference's own 512x512 screenshot.bmp (tests/golden)."""
    out = str(tmp_path / "shot512.bmp")
    r = subprocess.run([EXE, "200", "0", "512", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    raw = open(out, "rb").read()
    want = open(os.path.join(ROOT, "tests", "golden", "reference_screenshot_bmp_header.bin"), "rb").read()
    assert len(want) == 122 and len(raw) == 122 + 512 * 512 * 4
    assert raw[:122] == want
    px = np.frombuffer(raw, dtype="<u4", offset=122)
    assert (px >> 24 == 128).all()          # PutPixelSDL's alpha (SDLauxiliary.h:160)


def test_heterogeneous_run_through_reference_named_functions(tmp_path):
    """HeterogeneousRun: CreateSurface, GRID moved by the caller, CastPhotonBeams, DrawHet."""
    import raytracer_amd as pb
    out = str(tmp_path / "het.bmp")
    r = subprocess.run([EXE, "800", "1", "128", out, "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout
    assert "Creating grid" in r.stdout and "Minimum value:" in r.stdout
    img = read_bmp32(out)
    kw = dict(photon_number=800, bounces=1, screen_width=128, screen_height=128, heterogeneous=1, one_dimensional=1)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.set_grid(*pb.make_grid())
        n_beams, n_segs = ctx.cast_photon_beams()
        ref = pb.quantise_argb(ctx.draw(mode=pb.MODE_73HET, cull="bins"))
    assert f"Beams size: {n_beams}" in r.stdout and f"Segment size: {n_segs}" in r.stdout
    assert np.array_equal(img, ref)
    assert (img & 0xFF).max() > 20


def test_stop_motion_through_reference_named_functions(tmp_path):
    """ProduceStopMotion (skeleton.cpp:265-328): three frames, the medium replaced and the beams re-cast between
    them.  The first frame is the plain heterogeneous run; later frames keep the first beam list and differ."""
    exe = os.path.join(ROOT, "examples", "stop_motion_headless")
    assert os.path.exists(exe), "examples/stop_motion_headless is not built (make -C raytracer_amd/csrc)"
    prefix = str(tmp_path / "shot_")
    r = subprocess.run([exe, "600", "128", "16", "19", prefix, "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=180)
    assert r.returncode == 0, r.stdout
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("frame ")]
    assert len(lines) == 3
    imgs = [read_bmp32(f"{prefix}{i}_.bmp") for i in (16, 17, 18)]
    assert all(im.shape == (128, 128) and (im & 0xFF).max() > 10 for im in imgs)
    assert not np.array_equal(imgs[0], imgs[1]) and not np.array_equal(imgs[1], imgs[2])
    import raytracer_amd as pb
    kw = dict(photon_number=600, screen_width=128, screen_height=128, heterogeneous=1, one_dimensional=1)
    with pb.Context(pb.control_defaults(**kw)) as ctx:
        ctx.set_scene()
        ctx.set_grid(*pb.make_grid(time=0.016))
        ctx.cast_photon_beams()
        ref0 = pb.quantise_argb(ctx.draw(mode=pb.MODE_73HET, cull="bins"))
    assert np.array_equal(imgs[0], ref0)


def test_run_analysis_writes_the_reference_csv_files(tmp_path):
    """RunAnalysis (Analysis.h:81-122) on the GPU, through the host layer: NSD.csv as the reference has it
    committed and rRMSE.csv as the reference's own function writes it.  The device's exp/pow may differ from
    the host's in the last bit, which six printed decimals show only rarely: cells are compared as text, with
    at most 1 % allowed to differ, and then by no more than one unit of the last printed digit."""
    exe = os.path.join(ROOT, "examples", "skeleton_headless")
    r = subprocess.run([exe, "analysis"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    gold = os.path.join(ROOT, "tests", "golden")
    for mine, ref, rows, cols in (("NSD.csv", "reference_NSD.csv", 250, 9), ("rRMSE.csv", "ref_rRMSE.csv", 317, 4)):
        a = open(tmp_path / mine).read().splitlines()
        b = open(os.path.join(gold, ref)).read().splitlines()
        assert len(a) == len(b) == rows
        cells = differ = 0
        for la, lb in zip(a, b):
            ca, cb = la.split(","), lb.split(",")
            assert len(ca) == len(cb) == cols and ca[0] == cb[0]
            for x, y in zip(ca[1:], cb[1:]):
                cells += 1
                if x != y:
                    differ += 1
                    assert x and y and abs(float(x) - float(y)) <= 1.5e-6 * max(1.0, abs(float(y))), (mine, la, lb)
        assert differ <= 0.01 * cells, (mine, differ, cells)


def test_run_analysis_matches_oracle_values():
    import orc
    import raytracer_amd as pb
    with pb.Context(pb.control_defaults()) as ctx:
        for table in (0, 1):
            fc, v, p = ctx.run_analysis(table)
            ofc, ov, op = orc.run_analysis(table)
            assert np.array_equal(fc, ofc) and np.array_equal(p, op)
            ok = np.isfinite(ov) & (p == 1)
            np.testing.assert_allclose(v[ok], ov[ok], rtol=3e-7)      # sqrtf rounds both to float
            assert np.array_equal(np.isnan(v), np.isnan(ov))


def test_dumped_segment_set_renders_the_reference_screenshot(tmp_path):
    """examples/skeleton_headless segments <file>: LoadTestModel, TransformationMatrix, LoadSegments, DrawStandard,
    SDL_SaveImage over tests/golden/cfg1_tree_segments.bin (the reference's own tree segments at Control.h as
    committed) gives the reference's committed screenshot to 1/255 on at most 0.1 % of the pixels - no Python in
    the loop.  pb_save_segments / pb_load_segments round-trip the set on the GPU."""
    from PIL import Image
    import raytracer_amd as pb
    fixture = os.path.join(ROOT, "tests", "golden", "cfg1_tree_segments.bin")
    out = str(tmp_path / "ref.bmp")
    r = subprocess.run([EXE, "segments", fixture, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    img = read_bmp32(out)
    shot = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "reference_screenshot.png")).convert("RGBA")).astype(np.int32)
    mine = np.stack([(img >> 16) & 255, (img >> 8) & 255, img & 255], -1).astype(np.int32)
    d = np.abs(mine - shot[..., :3]).max(-1)
    assert d.max() <= 1 and (d > 0).mean() <= 1e-3, (d.max(), int((d > 0).sum()))
    with pb.Context(pb.control_defaults()) as ctx:
        ctx.set_scene()
        ctx.load_segments(fixture)
        assert ctx.segment_count() == 987
        a = ctx.draw(cull="bins")
        assert np.array_equal(pb.quantise_argb(a), img)
        ctx.cast_photon_beams(300)
        p = str(tmp_path / "dump.bin")
        ctx.save_segments(p)
        segs = ctx.download_segments()
        lo, hi = ctx.root_box()
        b = ctx.draw(cull="bins")
        s2, l2, h2 = pb.read_segments_file(p)
        assert s2.tobytes() == segs.tobytes() and np.array_equal(l2, lo) and np.array_equal(h2, hi)
        ctx.load_segments(fixture)
        ctx.load_segments(p)
        assert np.array_equal(ctx.draw(cull="bins"), b)


@pytest.mark.parametrize("args", [["1500", "2", "200"], ["800", "1", "136", None, "1"]],
                         ids=["homogeneous", "heterogeneous"])
def test_frame_spread_over_several_contexts_is_the_same_image(tmp_path, args):
    """pbhost::devices: the frame's rows go round the listed devices in bands of 64, one host thread and one
    context per entry.  On a one-GPU box the list names GPU 0 two and three times (each entry still gets its own
    context, scene, grid and segment upload); the BMP must equal the single-context one byte for byte - also for
    a height that is not a multiple of the band."""
    assert os.path.exists(EXE), "examples/skeleton_headless is not built (make -C raytracer_amd/csrc)"
    shots = {}
    for name, dev in (("one", None), ("two", "0,0"), ("three", "0,0,0")):
        out = str(tmp_path / f"{name}.bmp")
        argv = [EXE] + [out if a is None else a for a in (args + [None] if len(args) == 3 else args)]
        env = dict(os.environ)
        env.pop("PB_DEVICES", None)
        if dev:
            env["PB_DEVICES"] = dev
        r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=180, env=env)
        assert r.returncode == 0, r.stdout
        shots[name] = open(out, "rb").read()
    assert len(shots["one"]) > 122 and shots["two"] == shots["one"] and shots["three"] == shots["one"]
    img = read_bmp32(str(tmp_path / "one.bmp"))
    assert (img & 0xFF).max() > 50


def test_device_list_errors_are_reported(tmp_path):
    env = dict(os.environ, PB_DEVICES="0,x")
    r = subprocess.run([EXE, "100", "0", "64", str(tmp_path / "a.bmp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=60, env=env)
    assert r.returncode == 1 and "PB_DEVICES" in r.stdout
    env = dict(os.environ, PB_DEVICES="0,63")
    r = subprocess.run([EXE, "100", "0", "64", str(tmp_path / "b.bmp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=60, env=env)
    assert r.returncode == 1 and "error" in r.stdout


def test_keyboard_loop_without_a_window(tmp_path):
    """skeleton.cpp:226-230, :475-486, :534-610 through the host layer (examples/keys_headless.cpp): Update / UserInput
    fed from a script of key presses.  A camera key moves the walls of the next frame while the beams stay as cast
    (DrawStandard makes its matrix from the globals every frame, skeleton.cpp:333-334), `r` restores camera and
    rotation and sets the light where the reference's reset puts it, a light key shows only after a new photon pass."""
    import raytracer_amd as pb
    exe = os.path.join(ROOT, "examples", "keys_headless")
    assert os.path.exists(exe), "examples/keys_headless is not built (make -C raytracer_amd/csrc)"
    prefix = str(tmp_path / "k")
    keys = ["j", "w", "left", "r", "l", "cast"]
    r = subprocess.run([exe, "600", "128", prefix] + keys, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("Render time:") == len(keys) + 1       # one Update per frame and the one that ends the loop
    frames = [read_bmp32(f"{prefix}_{k}.bmp") for k in range(len(keys))]
    # the state the reference's switch leaves: reset, then light.x += 0.1 (float arithmetic through double, as there)
    assert "camera 0 0 -3 theta 0 0 0 light 0.1 -0.5 -0.7" in r.stdout
    kw = dict(photon_number=600, screen_width=128, screen_height=128)
    ctl = pb.control_defaults(**kw)
    with pb.Context(ctl) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        plain = pb.quantise_argb(ctx.draw(cull="bins"))
        cam = np.float32(np.float64(np.float32(-3.0)) + 0.1)                       # camera.z += 0.1
        ctx.set_scene(matrix=pb.transformation_matrix([0.0, 0.0, 0.0], [0.0, 0.0, float(cam), 1.0]))
        moved = pb.quantise_argb(ctx.draw(cull="bins"))
        th = np.float32(np.float64(np.float32(0.0)) + 3.14159265 / 180)            # theta.y += PI / 180
        ctx.set_scene(matrix=pb.transformation_matrix([0.0, float(th), 0.0], [0.0, 0.0, float(cam), 1.0]))
        turned = pb.quantise_argb(ctx.draw(cull="bins"))
    assert np.array_equal(frames[0], plain)          # `j` moved the light only: nothing of it shows before a new photon pass
    assert np.array_equal(frames[1], moved) and not np.array_equal(moved, plain)
    assert np.array_equal(frames[2], turned) and not np.array_equal(turned, moved)
    assert np.array_equal(frames[3], plain)          # reset
    assert np.array_equal(frames[4], plain)          # `l`: the light again
    # after the new photon pass the beams come from the moved light
    lp = [float(np.float32(np.float64(np.float32(0.0)) + 0.1)), -0.5, -0.7, 1.0]
    with pb.Context(pb.control_defaults(light_position=lp, **kw)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        relit = pb.quantise_argb(ctx.draw(cull="bins"))
    assert np.array_equal(frames[5], relit) and not np.array_equal(relit, plain)
