"""The device functions the kernels inline, one GPU lane per query, against tests/golden/ref_primitives.npz:
inputs and outputs of the REFERENCE's own ClosestIntersection, HitBoundingBox and HitCylinder (compiled from
its headers by oracle/gen_golden.py).  Bit-exact: these are the discontinuous predicates that decide which
(ray, segment) pairs contribute."""
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_primitives.npz")


@pytest.fixture(scope="module")
def pb():
    import raytracer_amd as pb
    if pb.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path cannot run (there is no CPU fallback)")
    return pb


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def ctx(pb):
    with pb.Context(pb.control_defaults()) as c:
        c.set_scene()
        yield c


def test_closest_intersection_matches_reference_vectors(ctx, gold):
    found, pos, dist, idx = ctx.probe_closest_intersection(gold["ci_start"], gold["ci_dir"])
    assert np.array_equal(found, gold["ci_found"])
    hit = gold["ci_found"] == 1
    assert hit.sum() > 1500
    assert np.array_equal(idx[hit], gold["ci_index"][hit])
    assert np.array_equal(dist[hit], gold["ci_dist"][hit])
    assert np.array_equal(pos[hit], gold["ci_pos"][hit])


def test_hit_bounding_box_matches_reference_vectors(ctx, gold):
    # the GPU form is specialised to rays from the camera at (0,0,0): the golden cases with that start
    cam = (gold["bb_start"][:, :3] == 0).all(1)
    assert cam.sum() >= 2000
    bmin, bmax = gold["bb_min"][cam], gold["bb_max"][cam]
    lo = np.stack([bmin[:, 0], bmin[:, 1], bmax[:, 2]], 1)      # un-flip: min_b = (min.x, min.y, max.z)
    hi = np.stack([bmax[:, 0], bmax[:, 1], bmin[:, 2]], 1)
    hit, dist = ctx.probe_hit_bounding_box(lo, hi, gold["bb_dir"][cam])
    assert np.array_equal(hit, gold["bb_hit"][cam])
    h = hit == 1
    assert 200 < h.sum() < cam.sum()
    # |hit - start| as a vec4 length with w = 1 - 1 = 0, float operations in glm's order
    p = gold["bb_point"][cam][h].astype(np.float32)
    want = np.sqrt((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + (p[:, 2] * p[:, 2] + np.float32(0)))
    assert want.dtype == np.float32 and np.array_equal(dist[h], want)


def _segments_from_gold(pb, gold, sel):
    n = int(sel.sum())
    segs = np.zeros(n, pb.SEGMENT_DTYPE)
    segs["start"] = gold["cy_s0"][sel][:, :3]; segs["end"] = gold["cy_s1"][sel][:, :3]
    segs["orig_start"] = gold["cy_so"][sel][:, :3]; segs["radius"] = gold["cy_radius"][sel]
    # a gate box that every ray passes: the probe reports the cylinder, the box is tested on its own above
    segs["cull_lo"] = -10.0; segs["cull_hi"] = 10.0
    segs["energy"] = 0.5; segs["seg_id"] = 1
    return segs


def test_hit_cylinder_matches_reference_vectors(pb, ctx, gold):
    cam = (gold["cy_start"][:, :3] == 0).all(1)
    assert cam.sum() > 5000
    segs = _segments_from_gold(pb, gold, cam)
    cyl, val, st = ctx.probe_pairs(segs, gold["cy_dir"][cam])
    want = gold["cy_out"][cam]
    valid = want[:, 1] == 1
    assert valid.sum() > 1000
    assert np.array_equal(cyl[:, 0], want[:, 0])          # HitCylinder's return value
    assert np.array_equal(cyl[:, 1], want[:, 1])          # CylIntersection.valid
    assert np.array_equal(st == 2, valid)
    assert np.array_equal(cyl[valid], want[valid])        # tb-, tb+, tc-, tc+, entry, exit: every bit
    # Integral_721 on those hits: the oracle's value (float exp; device and host expf differ in the last bit,
    # which the exp(x) - 1 cancellation magnifies to ~1e-4 relative at worst)
    d = gold["cy_dir"][cam][valid]
    s0, s1 = gold["cy_s0"][cam][valid], gold["cy_s1"][cam][valid]
    ext = np.float32(0.035) + np.float32(0.005)
    ref = np.array([orc.integral_721(s0[i], s1[i], want[valid][i, 2], want[valid][i, 4], want[valid][i, 5], ext, d[i])
                    for i in range(valid.sum())])
    ok = np.isfinite(ref)
    # (near-zero chords, tc- ~ tc+, are pure cancellation: absolute 2e-5 there)
    np.testing.assert_allclose(val[valid][ok], ref[ok], rtol=2e-4, atol=2e-5)
    # the default fast form: same hits up to grazing cases, same values to the stated tolerance
    cyl_f, val_f, st_f = ctx.probe_pairs(segs, gold["cy_dir"][cam], fast=True)
    assert ((st_f == 2) != valid).mean() < 0.01
    both = (st_f == 2) & valid & np.isfinite(val)
    # the exact-order value carries the reference's own float-exp cancellation noise (up to a few 1e-4
    # absolute on short chords), which the fast form's series for exp(x) - 1 does not have
    np.testing.assert_allclose(val_f[both], val[both], rtol=2e-3, atol=3e-4)
    rel = np.abs(val_f[both] - val[both]) / np.maximum(np.abs(val[both]), 1e-3)
    assert np.median(rel) < 2e-4 and rel.mean() < 2e-3


def test_hit_cone_matches_reference_vectors(pb, ctx, gold):
    """HitCone (Intersections.h:145-305) through the cone branch of the generic estimator: adaptive-width
    segments (flags bit 0), whole cones and frusta.  Integral_721_ada against the oracle's Riemann sum."""
    n = len(gold["co_out"])
    segs = np.zeros(n, pb.SEGMENT_DTYPE)
    segs["start"] = gold["co_s0"][:, :3]; segs["end"] = gold["co_s1"][:, :3]
    segs["orig_start"] = gold["co_so"][:, :3]; segs["radius"] = gold["co_radius"]
    segs["cull_lo"] = -10.0; segs["cull_hi"] = 10.0
    segs["energy"] = 0.5; segs["seg_id"] = 1; segs["flags"] = 1
    cyl, val, st = ctx.probe_pairs(segs, gold["co_dir"], mode=pb.MODE_GENERIC)
    want = gold["co_out"]
    valid = want[:, 1] == 1
    assert valid.sum() > 1000
    # cos/sin of the cone angle come from the device's double routines rounded to float: allow a handful of
    # segments whose angle sits on a rounding boundary, everything else is bit for bit
    same = (cyl.view(np.uint32) == want.view(np.uint32)).all(1)
    assert (~same).sum() <= 3, f"{(~same).sum()} of {n} HitCone results differ"
    assert np.array_equal(st[same] == 2, valid[same])
    idx = np.flatnonzero(valid & same)[:400]
    ext = np.float32(0.035) + np.float32(0.005)
    ref = np.array([orc.integral_721_ada(gold["co_s0"][i], gold["co_s1"][i], gold["co_so"][i], float(gold["co_radius"][i]),
                                         want[i, 2], want[i, 4], want[i, 5], ext, 0.005, gold["co_dir"][i]) for i in idx])
    fin = np.isfinite(ref)
    assert fin.sum() > 300
    np.testing.assert_allclose(val[idx][fin], ref[fin], rtol=1e-9)
