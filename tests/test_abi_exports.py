"""The C-ABI library loads without a GPU and exports every function include/photon_beams.h declares;
its host-only entry points (scene, matrix, surface) agree with the CPU oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np

import orc
import raytracer_amd as pb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "photon_beams.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 28 and "pb_render" in names and "pb_set_grid" in names and "pb_emit_beams" in names
    L = pb.library()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared but not exported: {missing}"


def test_struct_sizes_match_the_header():
    assert pb.SEGMENT_DTYPE.itemsize == 112 and pb.BEAM_DTYPE.itemsize == 80
    c = pb.control_defaults()
    assert (c.screen_width, c.photon_number, c.seed) == (512, 1000, 42)
    assert abs(c.absorption_c - 0.035) < 1e-7 and abs(c.scattering_c - 0.005) < 1e-7


def test_host_scene_and_matrix_match_oracle():
    tri, nrm = pb.load_test_model()
    otri, onrm = orc.load_test_model()
    assert np.array_equal(np.asarray(tri).reshape(30, 3, 3), orc.tri_xyz(otri).reshape(30, 3, 3))
    for theta in [(0, 0, 0), (0.1, -0.2, 0.3)]:
        assert np.array_equal(pb.transformation_matrix(theta, (0, 0, -3, 1)), orc.transformation_matrix(theta, (0, 0, -3, 1)))


def test_create_surface_matches_oracle():
    for tn, height, time in [(27, -3.0, 0.016), (5, 0.25, 1.5), (1, 0.0, 0.0)]:
        pts, side = pb.create_surface(tn, height, time)
        ref = np.zeros((side * side, 4), np.float32)
        n = orc.lib().orc_create_surface(C.c_int(tn), C.c_float(height), C.c_double(time),
                                         ref.ctypes.data_as(C.POINTER(C.c_float)))
        assert n == side * side
        assert np.array_equal(pts, ref)
    # HeterogeneousRun's transform of the lattice (skeleton.cpp:119-125)
    sigma, side = pb.make_grid()
    o = orc.Oracle(orc.control_defaults())
    osigma, oside = o.make_grid()
    assert side == oside == 28 and np.array_equal(sigma, osigma)
    assert sigma.min() >= 0.0 and sigma.max() > 0.01


def test_without_gpu_creation_fails_loudly():
    if pb.device_count() > 0:
        return
    try:
        pb.Context(pb.control_defaults())
    except pb.PhotonBeamsError as e:
        assert "no GPU" in str(e) or "no CPU path" in str(e)
    else:
        raise AssertionError("Context creation must fail without a GPU")
