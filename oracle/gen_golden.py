"""TEST INFRASTRUCTURE — writes tests/golden/ref_primitives.npz.

Runs the REFERENCE's own functions (oracle/_ref/libref.so, built by `make -C oracle ref`
from Source/Intersections.h, TestModelH.h and Tessendorf.h where they lie under
/root/reference) on seeded inputs and stores inputs and outputs.  Only works where the
reference tree exists; the committed .npz is what travels.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import orc  # noqa: E402


def camera_dir(rng, n, size=512):
    px = rng.integers(0, size, n).astype(np.float32) - np.float32(size / 2)
    py = rng.integers(0, size, n).astype(np.float32) - np.float32(size / 2)
    d = np.stack([px, py, np.full(n, size, np.float32), np.ones(n, np.float32)], 1)
    d = d / np.sqrt((d * d).sum(1, keepdims=True)).astype(np.float32)
    return d.astype(np.float32)


def main():
    assert orc.ref() is not None, "build oracle/_ref first: make -C oracle ref"
    rng = np.random.default_rng(20261111)
    out = {}
    tri, nrm = orc.load_test_model(use_ref=True)
    out["model_tri"], out["model_normals"] = tri, nrm
    out["surface_27_m3_0016"] = orc.create_surface(27, -3.0, 0.016, use_ref=True)
    out["surface_9_m3_0500"] = orc.create_surface(9, -3.0, 0.5, use_ref=True)
    m16 = orc.transformation_matrix()
    out["matrix"] = m16

    # realistic segments: the configuration-1 items of the restated emission (inputs only)
    o = orc.Oracle()
    o.cast_photon_beams(); o.bound_photon_beams(); o.build_tree()
    items = o.export_segments(tree=False)
    tree = o.export_segments(tree=True)

    # ClosestIntersection
    n = 2000
    start = np.zeros((n, 4), np.float32); start[:, 3] = 1
    dirs = camera_dir(rng, n)
    k = n // 2
    start[k:, :3] = rng.uniform(-0.95, 0.95, (n - k, 3)).astype(np.float32) + np.array([0, 0, 3], np.float32)
    rd = rng.normal(size=(n - k, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True).astype(np.float32)
    dirs[k:, :3] = rd
    dirs[k:, 3] = rng.integers(0, 2, n - k).astype(np.float32)
    res = [orc.closest_intersection(start[i], dirs[i], m16, tri, use_ref=True) for i in range(n)]
    out["ci_start"], out["ci_dir"] = start, dirs
    out["ci_found"] = np.array([r[0] for r in res], np.uint8)
    out["ci_pos"] = np.stack([r[1] for r in res]); out["ci_dist"] = np.array([r[2] for r in res], np.float32)
    out["ci_index"] = np.array([r[3] for r in res], np.int32)

    # HitBoundingBox (boxes in the reference's flipped-z storage)
    n = 4000
    bmin = np.ones((n, 4), np.float32); bmax = np.ones((n, 4), np.float32)
    pick = tree[rng.integers(0, len(tree), n)]
    bmin[:, 0] = pick["cull_lo"][:, 0]; bmin[:, 1] = pick["cull_lo"][:, 1]; bmin[:, 2] = pick["cull_hi"][:, 2]
    bmax[:, 0] = pick["cull_hi"][:, 0]; bmax[:, 1] = pick["cull_hi"][:, 1]; bmax[:, 2] = pick["cull_lo"][:, 2]
    start = np.zeros((n, 4), np.float32); start[:, 3] = 1
    dirs = camera_dir(rng, n)
    q = n // 2
    start[q:, :3] = rng.uniform(-1.5, 1.5, (n - q, 3)).astype(np.float32) + np.array([0, 0, 2.5], np.float32)
    rd = rng.normal(size=(n - q, 3)).astype(np.float32)
    rd[rng.random((n - q, 3)) < 0.1] = 0          # axis-parallel directions
    rd[(rd == 0).all(1)] = np.array([0, 0, 1], np.float32)
    dirs[q:, :3] = rd; dirs[q:, 3] = rng.integers(0, 2, n - q).astype(np.float32)
    res = [orc.hit_bounding_box(bmin[i], bmax[i], start[i], dirs[i], use_ref=True) for i in range(n)]
    out["bb_min"], out["bb_max"], out["bb_start"], out["bb_dir"] = bmin, bmax, start, dirs
    out["bb_hit"] = np.array([r[0] for r in res], np.uint8)
    out["bb_point"] = np.stack([np.where(r[0], r[1], 0).astype(np.float32) for r in res])

    # HitCylinder: rays from the camera aimed near the axis, the caps and past them
    n = 6000
    pick = items[rng.integers(0, len(items), n)]
    s0 = np.ones((n, 4), np.float32); s1 = np.ones((n, 4), np.float32); so = np.ones((n, 4), np.float32)
    s0[:, :3] = pick["start"]; s1[:, :3] = pick["end"]; so[:, :3] = pick["orig_start"]
    rad = pick["radius"].astype(np.float32)
    u = rng.uniform(-0.15, 1.15, (n, 1)).astype(np.float32)
    u[rng.random(n) < 0.2] = 0.0
    u[rng.random(n) < 0.2] = 1.0
    target = s0[:, :3] + u * (s1[:, :3] - s0[:, :3]) + rng.normal(scale=0.02, size=(n, 3)).astype(np.float32)
    dirs = np.ones((n, 4), np.float32)
    dirs[:, :3] = target
    dirs = (dirs / np.sqrt((dirs * dirs).sum(1, keepdims=True))).astype(np.float32)
    start = np.zeros((n, 4), np.float32); start[:, 3] = 1
    inside = rng.random(n) < 0.05               # a few origins inside / beside the cylinder
    start[inside, :3] = target[inside] + rng.normal(scale=0.01, size=(int(inside.sum()), 3)).astype(np.float32)
    res = np.stack([orc.hit_cylinder(start[i], dirs[i], s0[i], s1[i], so[i], float(rad[i]), use_ref=True)
                    for i in range(n)])
    out["cy_start"], out["cy_dir"], out["cy_s0"], out["cy_s1"], out["cy_so"], out["cy_radius"] = \
        start, dirs, s0, s1, so, rad
    out["cy_out"] = res

    # findRotationMatrix
    n = 500
    c = rng.normal(size=(n, 3)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True).astype(np.float32)
    w = np.tile(np.array([0, 1, 0], np.float32), (n, 1))
    w[::7, 1] = np.float32(0.99999994)
    out["rot_c"], out["rot_w"] = c, w
    out["rot_m"] = np.stack([orc.find_rotation_matrix(c[i], w[i], use_ref=True) for i in range(n)])

    # HitCone: the adaptive-width beams of configuration 1 (one cone per beam, apex at the beam's start),
    # a share of them cut into a frustum (seg.start moved along the axis) as the heterogeneous split does
    oa = orc.Oracle(orc.control_defaults(adaptive=1))
    oa.cast_photon_beams(); oa.bound_photon_beams()
    cones = oa.export_segments(tree=False)
    cones = cones[(cones["flags"] & 1) == 1]
    n = 6000
    pick = cones[rng.integers(0, len(cones), n)]
    s0 = np.ones((n, 4), np.float32); s1 = np.ones((n, 4), np.float32); so = np.ones((n, 4), np.float32)
    s0[:, :3] = pick["start"]; s1[:, :3] = pick["end"]; so[:, :3] = pick["orig_start"]
    cut = rng.random(n) < 0.3
    f = rng.uniform(0.1, 0.8, (n, 1)).astype(np.float32)
    s0[cut, :3] = (so[:, :3] + f * (s1[:, :3] - so[:, :3]))[cut]
    rad = pick["radius"].astype(np.float32)
    u = rng.uniform(-0.15, 1.15, (n, 1)).astype(np.float32)
    u[rng.random(n) < 0.15] = 0.0
    u[rng.random(n) < 0.15] = 1.0
    spread = (rad[:, None] * np.clip(u, 0.05, 1.0)).astype(np.float32)
    target = so[:, :3] + u * (s1[:, :3] - so[:, :3]) + rng.normal(size=(n, 3)).astype(np.float32) * spread
    dirs = np.ones((n, 4), np.float32)
    dirs[:, :3] = target
    dirs = (dirs / np.sqrt((dirs * dirs).sum(1, keepdims=True))).astype(np.float32)
    start = np.zeros((n, 4), np.float32); start[:, 3] = 1
    res = np.stack([orc.hit_cone(start[i], dirs[i], s0[i], s1[i], so[i], float(rad[i]), use_ref=True)
                    for i in range(n)])
    out["co_start"], out["co_dir"], out["co_s0"], out["co_s1"], out["co_so"], out["co_radius"] = \
        start, dirs, s0, s1, so, rad
    out["co_out"] = res
    print("cone true", int(res[:, 0].sum()), "cone valid", int(res[:, 1].sum()))

    # RunAnalysis: the reference's own function writes NSD.csv (identical to the file it has committed, kept as
    # tests/golden/reference_NSD.csv) and rRMSE.csv (kept as tests/golden/ref_rRMSE.csv)
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            orc.ref().ref_run_analysis()
        finally:
            os.chdir(cwd)
        gold = os.path.join(orc.ROOT, "tests", "golden")
        assert open(os.path.join(tmp, "NSD.csv")).read() == open(os.path.join(gold, "reference_NSD.csv")).read()
        shutil.copy(os.path.join(tmp, "rRMSE.csv"), os.path.join(gold, "ref_rRMSE.csv"))

    dst = os.path.join(orc.ROOT, "tests", "golden", "ref_primitives.npz")
    np.savez_compressed(dst, **out)
    cy = out["cy_out"]
    print("wrote", dst, os.path.getsize(dst), "bytes;",
          "ci found", int(out["ci_found"].sum()), "bb hit", int(out["bb_hit"].sum()),
          "cyl true", int(cy[:, 0].sum()), "cyl valid", int(cy[:, 1].sum()))


if __name__ == "__main__":
    main()
