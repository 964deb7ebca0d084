"""GPU LBVH (Morton keys, LDS radix sort, Karras hierarchy, refit) against its CPU twin: bit-exact keys,
permutation, topology and boxes; and the LBVH-driven frame against the scan-everything frame."""
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


def check_tree(nodes, n):
    """Every leaf and every inner node is reachable exactly once from node 0."""
    seen_leaf = np.zeros(n, bool); seen_node = np.zeros(n - 1, bool)
    stack = [0]
    while stack:
        v = stack.pop()
        assert not seen_node[v]; seen_node[v] = True
        for c in (int(nodes["left"][v]), int(nodes["right"][v])):
            if c >= 0:
                stack.append(c)
            else:
                assert not seen_leaf[~c]; seen_leaf[~c] = True
    assert seen_leaf.all() and seen_node.all()


@pytest.mark.parametrize("n_photons,bounces", [(1000, 0), (20000, 3)])
def test_lbvh_bit_exact_vs_cpu_twin(pb, n_photons, bounces):
    with pb.Context(pb.control_defaults(photon_number=n_photons, bounces=bounces)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        segs = ctx.download_segments()
        ctx.build_tree()
        keys, perm, nodes = ctx.download_lbvh()
    okeys, operm, onodes = orc.lbvh(segs)
    n = len(segs)
    assert (np.diff(keys.astype(np.uint64)) >= 0).all() if n > 1 else True
    assert np.array_equal(keys, okeys)
    assert np.array_equal(perm, operm)                    # stable: equal keys keep their input order
    assert sorted(perm.tolist()) == list(range(n))
    assert np.array_equal(nodes["left"], onodes["left"]) and np.array_equal(nodes["right"], onodes["right"])
    assert np.array_equal(nodes["lo"], onodes["lo"]) and np.array_equal(nodes["hi"], onodes["hi"])
    check_tree(nodes, n)


def test_lbvh_duplicate_keys_and_tiny_sets(pb):
    """All segments identical (every Morton key equal: the hierarchy rests on the position tiebreak), and
    sets of 1, 2 and 3 segments."""
    ctl = pb.control_defaults(screen_width=64, screen_height=64)
    with pb.Context(ctl) as ctx:
        ctx.set_scene()
        base = np.zeros(1, pb.SEGMENT_DTYPE)
        base["start"] = [0.0, -0.8, 2.5]; base["end"] = [0.1, 0.9, 2.7]; base["orig_start"] = base["start"]
        base["radius"] = 0.025; base["cull_lo"] = [-1, -1, 1]; base["cull_hi"] = [1, 1, 4]; base["energy"] = 0.5
        for n in (1, 2, 3, 200):
            segs = np.repeat(base, n)
            ctx.upload_segments(segs, [-1, -1, 1], [1, 1, 4])
            ctx.build_tree()
            a = ctx.draw(use_lbvh=True); b = ctx.draw(use_lbvh=False)
            assert np.array_equal(a, b), n
            assert a.max() > 0
            if n >= 2:
                keys, perm, nodes = ctx.download_lbvh()
                assert (keys == keys[0]).all() and np.array_equal(perm, np.arange(n))
                check_tree(nodes, n)
                ok, op, on = orc.lbvh(segs)
                assert np.array_equal(nodes["left"], on["left"]) and np.array_equal(nodes["right"], on["right"])


def test_lbvh_frame_equals_scan_frame(pb):
    """The acceleration structure only selects candidates: with it and without it the same pairs contribute
    (same counters) and the radiance agrees to the last bits of the differently ordered sums."""
    with pb.Context(pb.control_defaults(photon_number=10000, bounces=3)) as ctx:
        ctx.set_scene()
        ctx.cast_photon_beams()
        with pytest.raises(pb.PhotonBeamsError):
            ctx.draw(use_lbvh=True)                 # not built yet
        ctx.build_tree()
        ctx.enable_counters(True)
        ctx.set_precise(True)
        a = ctx.draw(0, 128, 512, 256, use_lbvh=True); ca = ctx.counters()
        b = ctx.draw(0, 128, 512, 256, use_lbvh=False); cb = ctx.counters()
    assert ca["valid_hits"] == cb["valid_hits"] and ca["cyl_hits"] == cb["cyl_hits"]
    assert ca["candidates"] == cb["candidates"]
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)
    assert a.max() > 1
