"""The CPU oracle against the reference: golden vectors produced by the reference's own
functions (tests/golden/ref_primitives.npz, written by oracle/gen_golden.py), the
reference's committed render (tests/golden/reference_screenshot.png, a copy of the
reference's screenshot.png) and the counts the reference prints for configuration 1."""
import os

import numpy as np
import pytest

import orc


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what):
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b)) | ((a == 0) & (b == 0))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} values differ; first at {np.argwhere(~same)[0]}"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_primitives.npz"))


def test_load_test_model_bit_exact(gold):
    tri, nrm = orc.load_test_model()
    assert tri.shape == (30, 12)
    assert_bit_equal(tri, gold["model_tri"], "LoadTestModel vertices")
    assert_bit_equal(nrm, gold["model_normals"], "LoadTestModel normals")


def test_create_surface_bit_exact(gold):
    assert_bit_equal(orc.create_surface(27, -3.0, 0.016), gold["surface_27_m3_0016"], "CreateSurface(27,-3,0.016)")
    assert_bit_equal(orc.create_surface(9, -3.0, 0.5), gold["surface_9_m3_0500"], "CreateSurface(9,-3,0.5)")


def test_closest_intersection_bit_exact(gold):
    tri, m16 = gold["model_tri"], gold["matrix"]
    n = len(gold["ci_found"])
    found = np.zeros(n, np.uint8); pos = np.zeros((n, 4), np.float32)
    dist = np.zeros(n, np.float32); idx = np.zeros(n, np.int32)
    for i in range(n):
        f, p, d, k = orc.closest_intersection(gold["ci_start"][i], gold["ci_dir"][i], m16, tri)
        found[i], pos[i], dist[i], idx[i] = f, p, d, k
    assert (found == gold["ci_found"]).all()
    assert found.sum() > 1000 and (found == 0).sum() > 10
    assert (idx == gold["ci_index"]).all()
    assert_bit_equal(dist, gold["ci_dist"], "ClosestIntersection distance")
    assert_bit_equal(pos, gold["ci_pos"], "ClosestIntersection position")


def test_hit_bounding_box_bit_exact(gold):
    n = len(gold["bb_hit"])
    hit = np.zeros(n, np.uint8); pt = np.zeros((n, 4), np.float32)
    for i in range(n):
        h, p = orc.hit_bounding_box(gold["bb_min"][i], gold["bb_max"][i], gold["bb_start"][i], gold["bb_dir"][i])
        hit[i] = h
        pt[i] = p if h else 0
    assert (hit == gold["bb_hit"]).all()
    assert hit.sum() > 300 and (hit == 0).sum() > 300
    assert_bit_equal(pt, gold["bb_point"], "HitBoundingBox entry point")


def test_hit_cylinder_bit_exact(gold):
    n = len(gold["cy_out"])
    out = np.stack([orc.hit_cylinder(gold["cy_start"][i], gold["cy_dir"][i], gold["cy_s0"][i], gold["cy_s1"][i],
                                     gold["cy_so"][i], float(gold["cy_radius"][i])) for i in range(n)])
    ref = gold["cy_out"]
    assert (out[:, 0] == ref[:, 0]).all() and (out[:, 1] == ref[:, 1]).all()
    # every branch class is present: miss, true+valid, true+invalid
    assert (ref[:, 0] == 0).sum() > 500 and (ref[:, 1] == 1).sum() > 1000
    assert ((ref[:, 0] == 1) & (ref[:, 1] == 0)).sum() >= 1
    assert_bit_equal(out, ref, "HitCylinder outputs")


def test_hit_cone_bit_exact(gold):
    """HitCone (Intersections.h:145-305) on adaptive-width beams of configuration 1, whole cones and frusta."""
    n = len(gold["co_out"])
    out = np.stack([orc.hit_cone(gold["co_start"][i], gold["co_dir"][i], gold["co_s0"][i], gold["co_s1"][i],
                                 gold["co_so"][i], float(gold["co_radius"][i])) for i in range(n)])
    ref = gold["co_out"]
    assert (ref[:, 0] == 0).sum() > 500 and (ref[:, 1] == 1).sum() > 1000
    assert ((ref[:, 0] == 1) & (ref[:, 1] == 0)).sum() >= 1
    assert_bit_equal(out, ref, "HitCone outputs")


def test_integral_721_ada_is_the_riemann_sum():
    """Integral_721_ada (BeamRadiance.h:63-84) is restated only (its header needs SDL): check the sum against
    an independent numpy evaluation of the same formula."""
    s0 = np.array([0.1, -0.9, 2.2, 1], np.float32); s1 = np.array([0.3, 0.8, 2.9, 1], np.float32)
    d = np.array([0.05, 0.1, 1, 1], np.float32); d /= np.float32(np.sqrt((d * d).sum()))
    tb, tc0, tc1, ext, sca, rad = 0.9, 2.4, 2.4731, 0.04, 0.005, 0.045
    got = orc.integral_721_ada(s0, s1, s0, rad, tb, tc0, tc1, ext, sca, d)
    bd = (s1[:3] - s0[:3]).astype(np.float64); bd /= np.linalg.norm(bd)
    cd = d[:3].astype(np.float64); cd /= np.linalg.norm(cd)
    cos = abs(float(np.float32(-(bd * cd).sum())))
    L = float(np.float32(np.linalg.norm((s0 - s1).astype(np.float64))))
    tc = np.float64(np.float32(tc0)) + 0.001 * np.arange(int(np.ceil((np.float32(tc1) - np.float32(tc0)) / 0.001)) + 2)
    tc = tc[tc < np.float64(np.float32(tc1))]
    tbs = np.float64(np.float32(tb)) - cos * (tc - np.float64(np.float32(tc0)))
    r = (np.float64(np.float32(rad)) / L) * tbs
    e = np.float64(np.float32(ext))
    want = (np.exp(-e * tc) * np.exp(-e * tbs) * (np.float64(np.float32(sca)) / r ** 2) * 0.001).sum()
    assert len(tc) in (73, 74) and abs(got - want) <= 1e-6 * want


def test_find_rotation_matrix_bit_exact(gold):
    n = len(gold["rot_c"])
    m = np.stack([orc.find_rotation_matrix(gold["rot_c"][i], gold["rot_w"][i]) for i in range(n)])
    assert_bit_equal(m, gold["rot_m"], "findRotationMatrix")


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref (the reference's own functions) is not built here")
def test_primitives_live_against_reference():
    """Fresh seeds against the reference's functions themselves (only where /root/reference exists)."""
    rng = np.random.default_rng(7)
    tri, _ = orc.load_test_model(use_ref=True)
    m16 = orc.transformation_matrix()
    for _ in range(300):
        d = np.append(rng.normal(size=3), 1.0).astype(np.float32)
        d /= np.float32(np.sqrt((d * d).sum()))
        s = np.array([0, 0, 0, 1], np.float32)
        a = orc.closest_intersection(s, d, m16, tri); b = orc.closest_intersection(s, d, m16, tri, use_ref=True)
        assert a[0] == b[0] and a[3] == b[3] and bits(a[2]) == bits(b[2])
        s0 = np.append(rng.uniform(-1, 1, 3) + [0, 0, 3], 1).astype(np.float32)
        s1 = np.append(rng.uniform(-1, 1, 3) + [0, 0, 3], 1).astype(np.float32)
        t = s0[:3] + rng.random() * (s1[:3] - s0[:3]) + rng.normal(scale=0.02, size=3)
        dd = np.append(t, 1.0).astype(np.float32); dd /= np.float32(np.sqrt((dd * dd).sum()))
        x = orc.hit_cylinder(s, dd, s0, s1, s0, 0.025); y = orc.hit_cylinder(s, dd, s0, s1, s0, 0.025, use_ref=True)
        assert_bit_equal(x, y, "HitCylinder live")
        lo = np.minimum(s0, s1); hi = np.maximum(s0, s1)
        bmin = np.array([lo[0], lo[1], hi[2], 1], np.float32); bmax = np.array([hi[0], hi[1], lo[2], 1], np.float32)
        p = orc.hit_bounding_box(bmin, bmax, s, dd); q = orc.hit_bounding_box(bmin, bmax, s, dd, use_ref=True)
        assert p[0] == q[0]
        if p[0]:
            assert_bit_equal(p[1], q[1], "HitBoundingBox live")


# ---------------------------------------------------------------- configuration 1
@pytest.fixture(scope="module")
def cfg1():
    o = orc.Oracle()
    n_beams = o.cast_photon_beams()
    n_items = o.bound_photon_beams()
    n_tree = o.build_tree()
    return o, n_beams, n_items, n_tree


def test_config1_counts_match_reference_run(cfg1):
    """957 beams (947 + 10), 1059 segments, 987 in the tree, 1705 nodes / 721 leaves / depth 49: the
    figures the unmodified reference prints and the survey measured for Control.h at HEAD."""
    o, n_beams, n_items, n_tree = cfg1
    assert (n_beams, o.beam_counts(), n_items, n_tree) == (957, (947, 10), 1059, 987)
    assert o.tree_stats() == {"nodes": 1705, "leaves": 721, "max_depth": 49}
    b = o.beams()
    np.testing.assert_allclose(b[0]["start"][:3], [0.02768737, -0.8003326, 2.503904], rtol=0, atol=1e-6)
    np.testing.assert_allclose(b[0]["end"][:3], [0.120168, 0.9999999, 2.196731], rtol=0, atol=1e-6)
    assert abs(float(b[0]["radius"]) - 0.02779691) < 1e-8
    lo, hi = o.root_box()
    np.testing.assert_allclose(lo, [-1, -1, 1.46026], atol=1e-5)
    np.testing.assert_allclose(hi, [1, 1, 3.64761], atol=1e-5)
    assert int((b["index_hit"][:947] == -1).sum()) == 116   # const beams leaving through the open front


@pytest.fixture(scope="module")
def cfg1_frame(cfg1):
    o = cfg1[0]
    return o.render_tree(0, 0, 512, 512)


def test_config1_image_matches_reference_screenshot(cfg1_frame, golden_dir):
    """Emission + segmentation + BuildTree + DrawStandard + PutPixelSDL reproduce the reference's own
    committed render: the reference rebuilt from source differs from it by 1/255 on 3 pixels."""
    from PIL import Image
    rgb, cnt = cfg1_frame
    argb = orc.put_pixels(rgb)
    ref = np.array(Image.open(os.path.join(golden_dir, "reference_screenshot.png")).convert("RGBA")).astype(np.int32)
    mine = np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], -1).astype(np.int32)
    d = np.abs(mine - ref[..., :3])
    assert d.max() <= 1
    assert int((d.max(-1) > 0).sum()) <= 8
    assert (argb >> 24 == 128).all()
    # per-ray work of the reference traversal (survey: 551.8 node visits, 182.2 HitCylinder calls, 11.99 valid)
    assert cnt["rays"] == 262144 and cnt["rays_hit_scene"] == 262052
    assert cnt["cyl_calls"] == 47768919 and cnt["cyl_true"] == 3145385 and cnt["valid_hits"] == 3144103


def test_flat_estimate_equals_tree_traversal(cfg1, cfg1_frame):
    """The form the GPU implements (root box and leaf-node box gate each segment) gives the very same
    doubles as the recursive traversal: same segments, same order."""
    o = cfg1[0]
    segs = o.export_segments(tree=True)
    lo, hi = o.root_box()
    tri, _ = orc.load_test_model()
    y0, h = 64, 96
    flat, cnt = orc.render_segments(o.control, orc.tri_xyz(tri), orc.transformation_matrix(), segs, lo, hi,
                                    0, y0, 512, h)
    assert np.array_equal(flat, cfg1_frame[0][y0:y0 + h])
    assert cnt["valid_hits"] > 100000


def test_run_analysis_reproduces_reference_csv(golden_dir):
    """RunAnalysis (Analysis.h:81-122): NSD.csv as the reference has it committed, rRMSE.csv as the reference's
    own function writes it here (oracle/gen_golden.py) — text for text."""
    nsd = orc.analysis_csv(*orc.run_analysis(0))
    rrmse = orc.analysis_csv(*orc.run_analysis(1))
    assert nsd == open(os.path.join(golden_dir, "reference_NSD.csv")).read()
    assert rrmse == open(os.path.join(golden_dir, "ref_rRMSE.csv")).read()
    assert nsd.count("\n") == 250 and rrmse.count("\n") == 317


def test_segment_file_fixture_is_the_config1_tree_set(cfg1, golden_dir, tmp_path):
    """tests/golden/cfg1_tree_segments.bin (written by tests/golden/make_cfg1_segments.py) holds exactly the
    oracle's configuration-1 tree segments and root box; the file functions of the C ABI (host code) round-trip
    a set byte for byte and refuse what is not a segment file."""
    import raytracer_amd as pb
    o = cfg1[0]
    segs, lo, hi = pb.read_segments_file(os.path.join(golden_dir, "cfg1_tree_segments.bin"))
    want = o.export_segments(tree=True)
    assert len(segs) == 987 and segs.tobytes() == want.tobytes()
    wlo, whi = o.root_box()
    assert np.array_equal(lo, wlo) and np.array_equal(hi, whi)
    raw = open(os.path.join(golden_dir, "cfg1_tree_segments.bin"), "rb").read()
    assert raw[:8] == b"PBSEGS\x00\x01" and len(raw) == 64 + 987 * 112
    p = str(tmp_path / "copy.bin")
    pb.write_segments_file(p, segs[:5], lo, hi)
    s2, l2, h2 = pb.read_segments_file(p)
    assert s2.tobytes() == segs[:5].tobytes() and np.array_equal(l2, lo) and np.array_equal(h2, hi)
    pb.write_segments_file(p, segs[:0], lo, hi)
    assert len(pb.read_segments_file(p)[0]) == 0
    open(p, "wb").write(b"not a segment file" * 10)
    with pytest.raises(pb.PhotonBeamsError):
        pb.read_segments_file(p)
    open(p, "wb").write(raw[:64 + 100])                       # truncated
    with pytest.raises(pb.PhotonBeamsError):
        pb.read_segments_file(p)
