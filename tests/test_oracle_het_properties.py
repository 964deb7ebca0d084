"""Extinction3D / FindAverageExtinction3D (Heterogeneous.h:29-185) have no output of the reference to be pinned by
(the header needs SDL to build, the reference holds no fixture for it): parity of the heterogeneous branch stays
"unpinned".  What can be checked without the reference are properties that follow from its text alone; the CPU
oracle's restatement (oracle/pb_oracle.cpp, the checker of the GPU walk) must have them:
  * the sample list is {0} + one sample per cell face crossed + {t_max}; a constant field has that constant as mean;
  * the first sample lies at distance 0, the last at t_max = |end - start| and is interpolated in the cell that holds
    the end point (the reference re-uses the index from before the step that overshot);
  * the interpolation reads b_r = b_l (missing braces at :158): along y it is linear, along x the slope is damped
    by (1 - fractional y) - visibly not the bilinear form."""
import numpy as np
import pytest

import orc

SIDE = 28
CELL = 2.0 / 27.0


def rays_inside(n, seed):
    rng = np.random.default_rng(seed)
    a = np.column_stack([rng.uniform(-0.98, 0.98, n), rng.uniform(-0.98, 0.98, n), rng.uniform(2.02, 3.98, n), np.ones(n)])
    b = np.column_stack([rng.uniform(-0.98, 0.98, n), rng.uniform(-0.98, 0.98, n), rng.uniform(2.02, 3.98, n), np.ones(n)])
    return a.astype(np.float32), b.astype(np.float32)


def crossings(a, b):
    """Cell faces strictly between the two end points: x and y faces every 2/27 from -1, z faces every 1 from 2."""
    n = 0
    for k, (origin, cell) in enumerate([(-1.0, CELL), (-1.0, CELL), (2.0, 1.0)]):
        ia = np.floor((np.float64(a[k]) - origin) / cell)
        ib = np.floor((np.float64(b[k]) - origin) / cell)
        n += int(abs(ib - ia))
    return n


def test_constant_field_mean_and_sample_count():
    sigma = np.full(SIDE * SIDE, 0.37, np.float32)
    a, b = rays_inside(300, 1)
    for s, e in zip(a, b):
        mean, de = orc.extinction3d(sigma, SIDE, s, e)
        assert abs(float(mean) - 0.37) <= 1e-6
        assert np.all(np.abs(de[:, 1] - np.float32(0.37)) <= 1e-7)
        assert len(de) == crossings(s, e) + 2, (s, e, len(de))


def test_first_sample_at_zero_last_at_t_max_distances_increase():
    sigma = np.random.default_rng(2).uniform(0.01, 0.7, SIDE * SIDE).astype(np.float32)
    a, b = rays_inside(300, 3)
    for s, e in zip(a, b):
        mean, de = orc.extinction3d(sigma, SIDE, s, e)
        t_max = np.float32(np.sqrt(np.sum((e[:3].astype(np.float32) - s[:3].astype(np.float32)) ** 2, dtype=np.float32)))
        assert de[0, 0] == 0.0
        assert abs(float(de[-1, 0]) - float(t_max)) <= 2e-7 * max(1.0, float(t_max))
        assert np.all(np.diff(de[:-1, 0]) >= 0) and de[-2, 0] <= de[-1, 0]
        assert abs(float(mean) - float(de[:, 1].astype(np.float64).mean())) <= 1e-6


def _positions(s, e, dist):
    d = (e[:3].astype(np.float64) - s[:3].astype(np.float64))
    d /= np.linalg.norm(d)
    return s[:3].astype(np.float64)[None, :] + dist.astype(np.float64)[:, None] * d[None, :]


def test_field_varying_along_y_is_interpolated_linearly_also_at_the_end_point():
    g = np.random.default_rng(4).uniform(0.05, 0.6, SIDE)
    sigma = np.repeat(g, SIDE).astype(np.float32)             # sigma[row * side + col] = g[row]
    a, b = rays_inside(200, 5)
    for s, e in zip(a, b):
        _, de = orc.extinction3d(sigma, SIDE, s, e)
        pos = _positions(s, e, de[:, 0])
        y = (pos[:, 1] + 1.0) / CELL
        want = np.interp(y, np.arange(SIDE), g.astype(np.float32).astype(np.float64))
        # a sample on a face may be booked to either cell; a continuous piecewise-linear field does not care
        assert np.all(np.abs(de[:, 1] - want) <= 2e-5), np.abs(de[:, 1] - want).max()


def test_field_varying_along_x_shows_b_r_equals_b_l():
    h = np.random.default_rng(6).uniform(0.05, 0.6, SIDE).astype(np.float32).astype(np.float64)
    sigma = np.tile(h, SIDE).astype(np.float32)               # sigma[row * side + col] = h[col]
    a, b = rays_inside(200, 7)
    seen_difference = 0
    for s, e in zip(a, b):
        _, de = orc.extinction3d(sigma, SIDE, s, e)
        # interior samples only: on a face the cell (and with it the damping) is a matter of rounding
        pos = _positions(s, e, de[:, 0])[1:-1]
        got = de[1:-1, 1]
        if len(pos) < 3:
            continue
        # the sample at a crossing belongs to the cell just entered: take the mid-point rule to find it robustly
        mid = 0.5 * (_positions(s, e, de[1:-1, 0]) + _positions(s, e, np.append(de[2:-1, 0], de[-1, 0])))
        cx = np.clip(np.floor((mid[:, 0] + 1.0) / CELL), 0, SIDE - 2).astype(int)
        cy = np.clip(np.floor((mid[:, 1] + 1.0) / CELL), 0, SIDE - 2).astype(int)
        px = np.abs((pos[:, 0] + 1.0) - cx * CELL) / CELL
        py = np.abs((pos[:, 1] + 1.0) - cy * CELL) / CELL
        as_written = h[cx] + (h[cx + 1] - h[cx]) * (1.0 - py) * px        # b_r = b_l
        bilinear = h[cx] + (h[cx + 1] - h[cx]) * px
        ok = np.abs(got - as_written) <= 3e-5
        assert ok.mean() > 0.97, (ok.mean(), np.abs(got - as_written).max())      # ties at cell corners excepted
        seen_difference += int((np.abs(as_written - bilinear) > 1e-3).sum())
        assert np.abs(got - bilinear)[np.abs(as_written - bilinear) > 1e-3].min(initial=1.0) > 5e-4
    assert seen_difference > 500


def _het_set(field, n_photons=250):
    """An oracle-made heterogeneous segment set over `field` (a side x side array of sigma_t), its root box and controls."""
    kw = dict(photon_number=n_photons, bounces=1, screen_width=64, screen_height=64, heterogeneous=1, one_dimensional=1)
    ctl = orc.control_defaults(**kw)
    o = orc.Oracle(ctl)
    o.set_grid(field, SIDE)
    o.cast_photon_beams()
    o.bound_photon_beams()
    segs = o.export_segments(tree=False)
    lo, hi = o.root_box()
    return kw, ctl, segs, lo, hi


def test_integral_73het_in_a_constant_field_is_the_homogeneous_integral_73():
    """BeamRadiance.h:211-288 against :159-192: where the field equals sigma_t = absorption + scattering everywhere,
    every sample, every running mean and every interpolated end value is that constant, so Integral_73Het's
    transmittances and its scattering (beam extinction / sigma_t x sigma_s) are Integral_73's.  The restatement of
    the heterogeneous branch (which no output of the reference pins) must reduce to the homogeneous one."""
    base = orc.control_defaults()
    sigma_t = np.float32(base.absorption_c) + np.float32(base.scattering_c)
    field = np.full(SIDE * SIDE, sigma_t, np.float32)
    kw, ctl, segs, lo, hi = _het_set(field)
    assert len(segs) > 1000 and np.allclose(segs["c_ext"], sigma_t, rtol=1e-6) and np.allclose(segs["e_ext"], sigma_t, rtol=1e-6)
    tri, _ = orc.load_test_model()
    args = (orc.tri_xyz(tri), orc.transformation_matrix(), segs, lo, hi, 0, 0, 64, 64)
    het, c_het = orc.render_segments(ctl, *args, mode=orc.MODE_73HET, grid_sigma=field, side=SIDE)
    hom_ctl = orc.control_defaults(**dict(kw, heterogeneous=0))
    hom, c_hom = orc.render_segments(hom_ctl, *args, mode=orc.MODE_73)
    assert c_het["valid_hits"] == c_hom["valid_hits"] > 1000 and hom.max() > 0.01
    # equal up to the rounding of a mean of equal samples (sum / n) and of float products taken in another order
    assert np.abs(het - hom).max() <= 2e-6 * max(1.0, hom.max())


def test_integral_73het_is_linear_in_the_beam_energy_and_additive_over_segments():
    rng = np.random.default_rng(11)
    field = (0.2 + 0.3 * rng.random(SIDE * SIDE)).astype(np.float32)
    kw, ctl, segs, lo, hi = _het_set(field, n_photons=150)
    tri, _ = orc.load_test_model()
    def frame(s):
        return orc.render_segments(ctl, orc.tri_xyz(tri), orc.transformation_matrix(), s, lo, hi, 0, 0, 64, 64,
                                   mode=orc.MODE_73HET, grid_sigma=field, side=SIDE)[0]
    whole = frame(segs)
    twice = segs.copy(); twice["energy"] *= 2
    assert whole.max() > 0.01
    assert np.abs(frame(twice) - 2 * whole).max() <= 1e-12 * max(1.0, whole.max())
    assert np.abs(frame(segs[0::2]) + frame(segs[1::2]) - whole).max() <= 1e-9 * max(1.0, whole.max())
