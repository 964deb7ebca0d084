"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and, where it has been
built, of the wrappers around the reference's own functions (oracle/_ref/libref.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")

MODE_721 = 721
MODE_73HET = 73
MODE_73 = 730
MODE_GENERIC = 471


class Control(C.Structure):
    """pb_control of include/photon_beams.h (Control.h:4-50)."""
    _fields_ = [
        ("screen_width", C.c_int32), ("screen_height", C.c_int32), ("ssaa", C.c_int32),
        ("photon_number", C.c_int32), ("bounces", C.c_int32), ("adaptive", C.c_int32),
        ("short_beams", C.c_int32), ("short_view", C.c_int32), ("scatter", C.c_int32),
        ("one_dimensional", C.c_int32), ("fixed_radius", C.c_int32), ("heterogeneous", C.c_int32),
        ("radius", C.c_float), ("absorbed", C.c_float), ("seed", C.c_uint32),
        ("light_position", C.c_float * 4), ("light_power", C.c_float * 3),
        ("absorption_c", C.c_float), ("scattering_c", C.c_float),
        ("camera", C.c_float * 4), ("theta", C.c_float * 3),
    ]


def control_defaults(**over):
    """Control.h:4-50 defaults; keyword arguments override fields."""
    c = Control()
    c.screen_width = 512; c.screen_height = 512; c.ssaa = 1
    c.photon_number = 1000; c.bounces = 0; c.adaptive = 0
    c.short_beams = 0; c.short_view = 0; c.scatter = 1
    c.one_dimensional = 0; c.fixed_radius = 0; c.heterogeneous = 0
    c.radius = 0.01; c.absorbed = 0.0; c.seed = 42
    c.light_position[:] = [0.0, -0.8, -0.5, 1.0]
    c.light_power[:] = [0.5, 0.5, 0.5]
    c.absorption_c = 0.035; c.scattering_c = 0.005
    c.camera[:] = [0.0, 0.0, -3.0, 1.0]
    c.theta[:] = [0.0, 0.0, 0.0]
    for k, v in over.items():
        if isinstance(v, (list, tuple)):
            getattr(c, k)[:] = v
        else:
            setattr(c, k, v)
    return c


SEGMENT_DTYPE = np.dtype([
    ("start", "<f4", 3), ("end", "<f4", 3), ("orig_start", "<f4", 3), ("radius", "<f4"),
    ("cull_lo", "<f4", 3), ("cull_hi", "<f4", 3), ("energy", "<f4", 3),
    ("s_ext", "<f4"), ("e_ext", "<f4"), ("c_ext", "<f4"),
    ("seg_id", "<i4"), ("id", "<i4"), ("flags", "<u4"), ("reserved", "<u4", 3),
])
assert SEGMENT_DTYPE.itemsize == 112

BEAM_DTYPE = np.dtype([
    ("start", "<f4", 4), ("end", "<f4", 4), ("omega_u", "<f4", 4), ("radius", "<f4"),
    ("energy", "<f4", 3), ("index_hit", "<i4"), ("flags", "<u4"), ("photon", "<u4"), ("order", "<u4"),
])
assert BEAM_DTYPE.itemsize == 80

NODE_DTYPE = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("left", "<i4"), ("right", "<i4")])


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays", "rays_hit_scene", "node_visits", "cyl_calls", "cyl_true", "valid_hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/liboracle.so is not built: run `make -C oracle` "
                               "or __graft_entry__.build()")
        L = C.CDLL(ORACLE_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Control)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_integral_721.restype = C.c_double
        L.orc_integral_73het.restype = C.c_double
        L.orc_integral_721_ada.restype = C.c_double
        for name in ("orc_cast_photon_beams", "orc_beam_counts", "orc_get_beams", "orc_bound_photon_beams",
                     "orc_build_tree", "orc_tree_stats", "orc_root_box", "orc_export_segments", "orc_set_grid",
                     "orc_make_grid", "orc_render_tree"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


_ref = None


def ref_available():
    """Whether oracle/_ref is built (nothing is loaded by asking)."""
    return os.path.exists(REF_SO)


def ref():
    """The reference's own functions, or None where oracle/_ref is not built.  Loaded on first use only."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
    return _ref


# ----------------------------------------------------------------- primitives
def _load_test_model(L, prefix):
    tri = np.zeros((64, 12), np.float32)
    nrm = np.zeros((64, 4), np.float32)
    n = getattr(L, prefix + "load_test_model")(_fp(tri), _fp(nrm), 64)
    assert n > 0
    return tri[:n].copy(), nrm[:n].copy()


def load_test_model(use_ref=False):
    """LoadTestModel: (n,12) v0 v1 v2 as xyzw, (n,4) normals."""
    return _load_test_model(ref() if use_ref else lib(), "ref_" if use_ref else "orc_")


def tri_xyz(tri_xyzw):
    """(n,12) xyzw vertices -> (n,9) xyz."""
    return np.ascontiguousarray(tri_xyzw.reshape(-1, 3, 4)[:, :, :3].reshape(-1, 9))


def transformation_matrix(theta=(0, 0, 0), camera=(0, 0, -3, 1)):
    m = np.zeros(16, np.float32)
    lib().orc_transformation_matrix(_fp(_f32(theta, 3)), _fp(_f32(camera, 4)), _fp(m))
    return m


def closest_intersection(start, dir, m16, tri_xyzw, use_ref=False):
    L = ref() if use_ref else lib()
    f = getattr(L, ("ref_" if use_ref else "orc_") + "closest_intersection")
    pos = np.zeros(4, np.float32); dist = C.c_float(0); idx = C.c_int(0)
    t = _f32(tri_xyzw)
    found = f(_fp(_f32(start, 4)), _fp(_f32(dir, 4)), _fp(_f32(m16, 16)), _fp(t), C.c_int(t.size // 12),
              _fp(pos), C.byref(dist), C.byref(idx))
    return bool(found), pos, np.float32(dist.value), idx.value


def hit_bounding_box(bmin, bmax, start, dir, use_ref=False):
    L = ref() if use_ref else lib()
    f = getattr(L, ("ref_" if use_ref else "orc_") + "hit_bounding_box")
    hit = np.zeros(4, np.float32)
    r = f(_fp(_f32(bmin, 4)), _fp(_f32(bmax, 4)), _fp(_f32(start, 4)), _fp(_f32(dir, 4)), _fp(hit))
    return bool(r), hit


def hit_cylinder(start, dir, s0, s1, so, radius, use_ref=False):
    L = ref() if use_ref else lib()
    f = getattr(L, ("ref_" if use_ref else "orc_") + "hit_cylinder")
    out = np.zeros(12, np.float32)
    f(_fp(_f32(start, 4)), _fp(_f32(dir, 4)), _fp(_f32(s0, 4)), _fp(_f32(s1, 4)), _fp(_f32(so, 4)),
      C.c_float(radius), _fp(out))
    return out


def hit_cone(start, dir, s0, s1, so, radius, use_ref=False):
    L = ref() if use_ref else lib()
    f = getattr(L, ("ref_" if use_ref else "orc_") + "hit_cone")
    out = np.zeros(12, np.float32)
    f(_fp(_f32(start, 4)), _fp(_f32(dir, 4)), _fp(_f32(s0, 4)), _fp(_f32(s1, 4)), _fp(_f32(so, 4)),
      C.c_float(radius), _fp(out))
    return out


def integral_721_ada(s0, s1, so, radius, tb_minus, tc_minus, tc_plus, extinction, scattering, dir):
    return lib().orc_integral_721_ada(_fp(_f32(s0, 4)), _fp(_f32(s1, 4)), _fp(_f32(so, 4)), C.c_float(radius),
                                      C.c_float(tb_minus), C.c_float(tc_minus), C.c_float(tc_plus),
                                      C.c_float(extinction), C.c_float(scattering), _fp(_f32(dir, 4)))


def run_analysis(table, extinction=float(np.float32(0.04))):
    """RunAnalysis (Analysis.h:81-122) restated: (first column, values, present) of NSD.csv (table 0) or
    rRMSE.csv (table 1).  EXT_ANALYSIS is the float 0.04f widened (Control.h:21, Analysis.h:15)."""
    cols = 8 if table == 0 else 3
    L = lib()
    L.orc_run_analysis.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    cap = 4096
    fc = np.zeros(cap); v = np.zeros((cap, cols)); p = np.zeros((cap, cols), np.uint8)
    n = L.orc_run_analysis(extinction, table, fc.ctypes.data, v.ctypes.data, p.ctypes.data, cap)
    return fc[:n], v[:n], p[:n]


def analysis_csv(first_col, values, present):
    """The text RunAnalysis writes: `<<` of a double (%g) then `"," << to_string(x)` (%f) per cell, empty
    where the estimator's kernel misses (Analysis.h:93-102)."""
    return "".join("%g" % first_col[r] + "".join("," + ("%f" % values[r, k] if present[r, k] else "")
                                                 for k in range(values.shape[1])) + "\n" for r in range(len(first_col)))


def find_rotation_matrix(c, w, use_ref=False):
    L = ref() if use_ref else lib()
    f = getattr(L, ("ref_" if use_ref else "orc_") + "find_rotation_matrix")
    m = np.zeros(9, np.float32)
    f(_fp(_f32(c, 3)), _fp(_f32(w, 3)), _fp(m))
    return m


def integral_721(s0, s1, tb_minus, tc_minus, tc_plus, extinction, dir):
    return lib().orc_integral_721(_fp(_f32(s0, 4)), _fp(_f32(s1, 4)), C.c_float(tb_minus), C.c_float(tc_minus),
                                  C.c_float(tc_plus), C.c_float(extinction), _fp(_f32(dir, 4)))


def create_surface(triangle_number=27, height=-3.0, time=0.016, use_ref=False):
    L = ref() if use_ref else lib()
    f = getattr(L, ("ref_" if use_ref else "orc_") + "create_surface")
    n = (triangle_number + 1) ** 2
    pts = np.zeros((n, 4), np.float32)
    got = f(C.c_int(triangle_number), C.c_float(height), C.c_double(time), _fp(pts))
    assert got == n
    return pts


def extinction3d(sigma, side, start, end, capacity=4096):
    de = np.zeros((capacity, 2), np.float32)
    mean = C.c_float(0)
    s = _f32(sigma, side * side)
    n = lib().orc_extinction3d(_fp(s), C.c_int(side), _fp(_f32(start, 4)), _fp(_f32(end, 4)), _fp(de),
                               C.c_int(capacity), C.byref(mean))
    return np.float32(mean.value), de[:min(n, capacity)].copy()


def integral_73het(sigma, side, seg_record, entry, exit_, extinction_c, scattering_c, dir):
    s = _f32(sigma, side * side)
    rec = np.ascontiguousarray(seg_record)
    return lib().orc_integral_73het(_fp(s), C.c_int(side), rec.ctypes.data_as(C.c_void_p), _fp(_f32(entry, 3)),
                                    _fp(_f32(exit_, 3)), C.c_float(extinction_c), C.c_float(scattering_c),
                                    _fp(_f32(dir, 4)))


def put_pixels(rgb):
    """PutPixelSDL over an (...,3) float64 array -> uint32 ARGB."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.zeros(rgb.shape[:-1], np.uint32)
    flat = rgb.reshape(-1, 3)
    o = out.reshape(-1)
    v = np.float32(255) * flat.astype(np.float32)
    v = np.minimum(np.maximum(v, np.float32(0)), np.float32(255))
    q = v.astype(np.uint32)
    o[:] = (np.uint32(128) << np.uint32(24)) + (q[:, 0] << np.uint32(16)) + (q[:, 1] << np.uint32(8)) + q[:, 2]
    return out


# ------------------------------------------------------------------- pipeline
class Oracle:
    """The restated pipeline: CastPhotonBeams -> BoundPhotonBeams -> BuildTree -> Draw*."""

    def __init__(self, control=None):
        self.control = control if control is not None else control_defaults()
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create(C.byref(self.control)))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cast_photon_beams(self, number=None, rng_kind=0):
        n = self.control.photon_number if number is None else number
        return self.L.orc_cast_photon_beams(self.h, C.c_int(n), C.c_int(rng_kind))

    def recast_photon_beams(self, rng_kind=1, seed=42):
        self.L.orc_recast_photon_beams.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        return self.L.orc_recast_photon_beams(self.h, C.c_int(rng_kind), C.c_uint64(seed))

    def beam_counts(self):
        a = C.c_int(0); b = C.c_int(0)
        self.L.orc_beam_counts(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def beams(self):
        a, b = self.beam_counts()
        out = np.zeros(a + b, BEAM_DTYPE)
        n = self.L.orc_get_beams(self.h, out.ctypes.data_as(C.c_void_p), C.c_int(a + b))
        assert n == a + b
        return out

    def bound_photon_beams(self):
        self.n_items = self.L.orc_bound_photon_beams(self.h)
        return self.n_items

    def build_tree(self, max_depth=4096):
        self.n_tree = self.L.orc_build_tree(self.h, C.c_int(max_depth))
        return self.n_tree

    def tree_stats(self):
        a = C.c_int(0); b = C.c_int(0); c = C.c_int(0)
        self.L.orc_tree_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"nodes": a.value, "leaves": b.value, "max_depth": c.value}

    def lost_segments(self):
        return int(self.L.orc_lost_segments(self.h))

    def root_box(self):
        lo = np.zeros(3, np.float32); hi = np.zeros(3, np.float32)
        self.L.orc_root_box(self.h, _fp(lo), _fp(hi))
        return lo, hi

    def export_segments(self, tree):
        n = self.n_tree if tree else self.n_items
        out = np.zeros(n, SEGMENT_DTYPE)
        got = self.L.orc_export_segments(self.h, C.c_int(1 if tree else 0), out.ctypes.data_as(C.c_void_p),
                                         C.c_int(n))
        assert got == n, (got, n)
        return out

    def make_grid(self, triangle_number=27, height=-3.0, time=0.016):
        side = triangle_number + 1
        sigma = np.zeros(side * side, np.float32)
        got = self.L.orc_make_grid(self.h, C.c_int(triangle_number), C.c_float(height), C.c_double(time), _fp(sigma))
        assert got == side
        return sigma, side

    def set_grid(self, sigma, side):
        s = _f32(sigma, side * side)
        self.L.orc_set_grid(self.h, _fp(s), C.c_int(side))

    def render_tree(self, x0, y0, w, h, mode=MODE_721, threads=0):
        rgb = np.zeros((h, w, 3), np.float64)
        cnt = Counters()
        threads = threads or max_threads()
        r = self.L.orc_render_tree(self.h, C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), C.c_int(mode),
                                   rgb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.c_int(threads))
        assert r == 0
        return rgb, cnt.as_dict()


def max_threads():
    return int(lib().orc_max_threads())


def render_segments(control, tri_xyz9, m16, segs, root_lo, root_hi, x0, y0, w, h, mode=MODE_721,
                    grid_sigma=None, side=0, threads=0, step_x=1, step_y=1):
    """The flat estimate the GPU implements, on the CPU: (h,w,3) float64 and counters.
    With steps > 1 the pixels are (x0 + i*step_x, y0 + j*step_y)."""
    rgb = np.zeros((h, w, 3), np.float64)
    cnt = Counters()
    segs = np.ascontiguousarray(segs)
    assert segs.dtype == SEGMENT_DTYPE
    t = _f32(tri_xyz9)
    g = _f32(grid_sigma) if grid_sigma is not None else None
    threads = threads or max_threads()
    r = lib().orc_render_segments_strided(C.byref(control), _fp(t), C.c_int(t.size // 9), _fp(_f32(m16, 16)),
                                  _fp(g) if g is not None else None, C.c_int(side),
                                  segs.ctypes.data_as(C.c_void_p), C.c_int(len(segs)),
                                  _fp(_f32(root_lo, 3)), _fp(_f32(root_hi, 3)),
                                  C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), C.c_int(step_x), C.c_int(step_y),
                                  C.c_int(mode),
                                  rgb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.c_int(threads))
    assert r == 0, r
    return rgb, cnt.as_dict()


def lbvh(segs):
    """CPU twin of the GPU LBVH: sorted Morton keys, permutation, inner nodes (NODE_DTYPE)."""
    segs = np.ascontiguousarray(segs)
    n = len(segs)
    keys = np.zeros(n, np.uint64); perm = np.zeros(n, np.uint32); nodes = np.zeros(max(n - 1, 0), NODE_DTYPE)
    r = lib().orc_lbvh(segs.ctypes.data_as(C.c_void_p), C.c_int(n), keys.ctypes.data_as(C.c_void_p),
                       perm.ctypes.data_as(C.c_void_p), nodes.ctypes.data_as(C.c_void_p))
    assert r == (n if n >= 2 else 0), r
    return keys, perm, nodes
