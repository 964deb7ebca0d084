"""ctypes binding of libphoton_beams.so (C ABI: include/photon_beams.h).

The method names follow the reference's functions for the path
(skeleton.cpp / CastPhotons.h / BeamRadiance.h): see each docstring.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libphoton_beams.so")

MODE_721 = 721     # BeamRadianceStandard + Integral_721
MODE_73HET = 73    # BVHBeamRadianceHet + Integral_73Het
MODE_73 = 730      # BeamRadiance with ONE_DIMENSIONAL, homogeneous medium: Integral_73
MODE_DEFAULT = 0    # the estimator Control selects
MODE_GENERIC = 471  # BeamRadiance as written: cones (HitCone + Integral_721_ada) for adaptive-width segments


class PhotonBeamsError(RuntimeError):
    pass


class Control(C.Structure):
    """pb_control: run-time form of Source/Control.h:4-50."""
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


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_hit_scene", C.c_uint64), ("candidates", C.c_uint64),
                ("cyl_hits", C.c_uint64), ("valid_hits", C.c_uint64), ("nodes_visited", C.c_uint64), ("gate_passes", C.c_uint64), ("shortcut_wrong", C.c_uint64),
                ("ms_render", C.c_float), ("ms_build", C.c_float), ("ms_prepare", C.c_float),
                ("ms_emit", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


SEGMENT_DTYPE = np.dtype([
    ("start", "<f4", 3), ("end", "<f4", 3), ("orig_start", "<f4", 3), ("radius", "<f4"),
    ("cull_lo", "<f4", 3), ("cull_hi", "<f4", 3), ("energy", "<f4", 3),
    ("s_ext", "<f4"), ("e_ext", "<f4"), ("c_ext", "<f4"),
    ("seg_id", "<i4"), ("id", "<i4"), ("flags", "<u4"), ("reserved", "<u4", 3),
])
BEAM_DTYPE = np.dtype([
    ("start", "<f4", 4), ("end", "<f4", 4), ("omega_u", "<f4", 4), ("radius", "<f4"),
    ("energy", "<f4", 3), ("index_hit", "<i4"), ("flags", "<u4"), ("photon", "<u4"), ("order", "<u4"),
])
NODE_DTYPE = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("left", "<i4"), ("right", "<i4")])
assert SEGMENT_DTYPE.itemsize == 112 and BEAM_DTYPE.itemsize == 80 and NODE_DTYPE.itemsize == 32

_lib = None


def library_path():
    return _LIB_PATH


def source_stamp():
    """sha256 (first 16 hex digits) over the kernel sources in raytracer_amd/csrc and the Makefile that holds their
    compiler flags, in name order: names the build that a committed profile was measured on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(src, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def _preload_hip_runtime():
    """libphoton_beams.so carries no HIP runtime of its own: make the process's one visible first.
    In a process that can import torch that is the runtime bundled with torch (so that torch tensors,
    streams and torch.distributed/RCCL share one runtime with the kernels); otherwise /opt/rocm's."""
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return path
        except OSError:
            continue
    raise PhotonBeamsError("no HIP runtime (libamdhip64.so) could be loaded")


def library():
    """The loaded libphoton_beams.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise PhotonBeamsError(
                f"{_LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(_LIB_PATH)
        L.pb_last_error.restype = C.c_char_p
        L.pb_destroy.argtypes = [C.c_void_p]
        L.pb_destroy.restype = None
        L.pb_control_defaults.restype = None
        L.pb_transformation_matrix.restype = None
        L.pb_transform_points.restype = None
        L.pb_reload_tuning.restype = None
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = library().pb_last_error()
        raise PhotonBeamsError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


def control_defaults(**over):
    """Control.h:4-50 defaults; keyword arguments override fields."""
    c = Control()
    library().pb_control_defaults(C.byref(c))
    for k, v in over.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            getattr(c, k)[:] = list(v)
        else:
            setattr(c, k, v)
    return c


def reload_tuning():
    """Have the library read the PB_TUNE_* environment switches again (it reads them when a context is created)."""
    library().pb_reload_tuning()


def device_count():
    return int(library().pb_device_count())


def load_test_model():
    """LoadTestModel (TestModelH.h:45-214): (30,9) world-space triangles and (30,3) normals."""
    tri = np.zeros((30, 9), np.float32)
    nrm = np.zeros((30, 3), np.float32)
    n = library().pb_load_test_model(_fp(tri), _fp(nrm), C.c_int(30))
    if n != 30:
        raise PhotonBeamsError(f"pb_load_test_model returned {n}")
    return tri, nrm


def transformation_matrix(theta=(0.0, 0.0, 0.0), camera=(0.0, 0.0, -3.0, 1.0)):
    """TransformationMatrix (skeleton.cpp:512-532), 16 floats column-major."""
    m = np.zeros(16, np.float32)
    library().pb_transformation_matrix(_fp(_f32(theta, 3)), _fp(_f32(camera, 4)), _fp(m))
    return m


def create_surface(triangle_number=27, height=-3.0, time=0.016):
    """CreateSurface (Tessendorf.h:47-75): the (triangle_number+1)^2 lattice points as an (n, 4) array."""
    side = triangle_number + 1
    pts = np.zeros((side * side, 4), np.float32)
    n = library().pb_create_surface(C.c_int(triangle_number), C.c_float(height), C.c_double(time), _fp(pts),
                                    C.c_int(side * side))
    if n != side * side:
        raise PhotonBeamsError(f"pb_create_surface returned {n}")
    return pts, side


def make_grid(matrix=None, triangle_number=27, height=-3.0, time=0.016):
    """HeterogeneousRun's grid set-up (skeleton.cpp:113-125): CreateSurface, then every point moved into
    camera space; returns (sigma_t = z of the moved points, side)."""
    pts, side = create_surface(triangle_number, height, time)
    m = transformation_matrix() if matrix is None else _f32(matrix, 16)
    library().pb_transform_points(_fp(m), _fp(pts), C.c_int(len(pts)))
    return np.ascontiguousarray(pts[:, 2]), side


def quantise_argb(rgb):
    """PutPixelSDL (SDLauxiliary.h:149-161) over an (...,3) float array -> uint32 0x80RRGGBB."""
    a = _f32(rgb)
    out = np.zeros(a.shape[:-1], np.uint32)
    _check(library().pb_quantise_argb(_fp(a), C.c_int(out.size), out.ctypes.data_as(C.POINTER(C.c_uint32))),
           "pb_quantise_argb")
    return out


def write_segments_file(path, segs, root_lo, root_hi):
    """pb_write_segments_file: the segment set and its root box as a file (host code, no GPU)."""
    segs = np.ascontiguousarray(segs)
    if segs.dtype != SEGMENT_DTYPE:
        raise ValueError("segments must have SEGMENT_DTYPE")
    _check(library().pb_write_segments_file(os.fsencode(path), segs.ctypes.data_as(C.c_void_p), C.c_int(len(segs)),
                                            _fp(_f32(root_lo, 3)), _fp(_f32(root_hi, 3))), "pb_write_segments_file")


def read_segments_file(path):
    """pb_read_segments_file: (segments, root_lo, root_hi) (host code, no GPU)."""
    n = C.c_int(0)
    lo = np.zeros(3, np.float32); hi = np.zeros(3, np.float32)
    L = library()
    _check(L.pb_read_segments_file(os.fsencode(path), None, C.c_int(0), C.byref(n), _fp(lo), _fp(hi)), "pb_read_segments_file")
    segs = np.zeros(n.value, SEGMENT_DTYPE)
    _check(L.pb_read_segments_file(os.fsencode(path), segs.ctypes.data_as(C.c_void_p), C.c_int(n.value), C.byref(n), _fp(lo), _fp(hi)),
           "pb_read_segments_file")
    return segs, lo, hi


class Context:
    """One GPU's state for the path: scene, segment set, LBVH.  One per process and GPU."""

    def __init__(self, control=None, device=0):
        self.L = library()
        self.control = control if control is not None else control_defaults()
        h = C.c_void_p()
        _check(self.L.pb_create(C.byref(self.control), C.c_int(device), C.byref(h)), "pb_create")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.pb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene ---------------------------------------------------------------
    def set_scene(self, tri_xyz=None, matrix=None):
        """The (matrix, triangles) every ClosestIntersection call receives (Intersections.h:62-68).
        Defaults: LoadTestModel and TransformationMatrix of the context's Control."""
        if tri_xyz is None:
            tri_xyz, _ = load_test_model()
        if matrix is None:
            matrix = transformation_matrix(list(self.control.theta), list(self.control.camera))
        t = _f32(tri_xyz)
        _check(self.L.pb_set_scene(self.h, _fp(t), C.c_int(t.size // 9), _fp(_f32(matrix, 16))), "pb_set_scene")

    def set_grid(self, sigma_t, side):
        s = _f32(sigma_t, side * side)
        _check(self.L.pb_set_grid(self.h, _fp(s), C.c_int(side)), "pb_set_grid")

    # -- segments ------------------------------------------------------------
    def extinction3d(self, starts, ends, max_samples=0):
        """Extinction3D for rows of xyz start/end points: (means, counts[, samples (n, max_samples, 2)])."""
        a = np.ascontiguousarray(starts, np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(ends, np.float32).reshape(-1, 3)
        n = len(a)
        mean = np.zeros(n, np.float32); cnt = np.zeros(n, np.int32)
        samp = np.zeros((n, max_samples, 2), np.float32) if max_samples else None
        _check(self.L.pb_extinction3d(self.h, _fp(a), _fp(b), C.c_int(n), _fp(mean),
                                      cnt.ctypes.data_as(C.POINTER(C.c_int)),
                                      _fp(samp) if samp is not None else None, C.c_int(max_samples)),
               "pb_extinction3d")
        return (mean, cnt, samp) if max_samples else (mean, cnt)

    def run_analysis(self, table, extinction=float(np.float32(0.04))):
        """RunAnalysis (Analysis.h:81-122): (kernel_width / mfp, values, present) of NSD.csv (table 0, 8
        estimators) or rRMSE.csv (table 1, 3 estimators)."""
        cols = 8 if table == 0 else 3
        n = C.c_int(0)
        cap = 4096
        fc = np.zeros(cap); v = np.zeros((cap, cols)); p = np.zeros((cap, cols), np.uint8)
        _check(self.L.pb_run_analysis(self.h, C.c_double(extinction), C.c_int(table), fc.ctypes.data_as(C.c_void_p),
                                      v.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), C.c_int(cap),
                                      C.byref(n)), "pb_run_analysis")
        return fc[:n.value], v[:n.value], p[:n.value]

    # ---- known-answer probes: one GPU lane per query through the kernels' own device functions ----
    def probe_closest_intersection(self, start_xyzw, dir_xyzw):
        a = np.ascontiguousarray(start_xyzw, np.float32).reshape(-1, 4)
        d = np.ascontiguousarray(dir_xyzw, np.float32).reshape(-1, 4)
        n = len(a)
        found = np.zeros(n, np.uint8); pos = np.zeros((n, 4), np.float32)
        dist = np.zeros(n, np.float32); idx = np.zeros(n, np.int32)
        _check(self.L.pb_probe_closest_intersection(self.h, _fp(a), _fp(d), C.c_int(n), found.ctypes.data_as(C.c_void_p),
                                                    _fp(pos), _fp(dist), idx.ctypes.data_as(C.c_void_p)),
               "pb_probe_closest_intersection")
        return found, pos, dist, idx

    def probe_hit_bounding_box(self, lo_xyz, hi_xyz, dir_xyzw):
        lo = np.ascontiguousarray(lo_xyz, np.float32).reshape(-1, 3)
        hi = np.ascontiguousarray(hi_xyz, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dir_xyzw, np.float32).reshape(-1, 4)
        n = len(lo)
        hit = np.zeros(n, np.uint8); dist = np.zeros(n, np.float32)
        _check(self.L.pb_probe_hit_bounding_box(self.h, _fp(lo), _fp(hi), _fp(d), C.c_int(n),
                                                hit.ctypes.data_as(C.c_void_p), _fp(dist)), "pb_probe_hit_bounding_box")
        return hit, dist

    def probe_pairs(self, segs, dir_xyzw, mode=MODE_721, fast=False):
        segs = np.ascontiguousarray(segs)
        assert segs.dtype == SEGMENT_DTYPE
        d = np.ascontiguousarray(dir_xyzw, np.float32).reshape(-1, 4)
        n = len(segs)
        cyl = np.zeros((n, 12), np.float32); val = np.zeros(n, np.float64); st = np.zeros(n, np.int32)
        _check(self.L.pb_probe_pairs(self.h, segs.ctypes.data_as(C.c_void_p), _fp(d), C.c_int(n), C.c_int(mode),
                                     C.c_int(1 if fast else 0), _fp(cyl), val.ctypes.data_as(C.c_void_p),
                                     st.ctypes.data_as(C.c_void_p)), "pb_probe_pairs")
        return cyl, val, st

    def upload_segments(self, segs, root_lo, root_hi):
        """The segment set BuildTree would hold (CastPhotons.h:83, :808) and CastPhotonBeams' root box."""
        segs = np.ascontiguousarray(segs)
        if segs.dtype != SEGMENT_DTYPE:
            raise ValueError("segments must have SEGMENT_DTYPE")
        _check(self.L.pb_upload_segments(self.h, segs.ctypes.data_as(C.c_void_p), C.c_int(len(segs)),
                                         _fp(_f32(root_lo, 3)), _fp(_f32(root_hi, 3))), "pb_upload_segments")

    def upload_segments_device(self, d_ptr, n, root_lo, root_hi, stream=0):
        _check(self.L.pb_upload_segments_d(self.h, C.c_void_p(d_ptr), C.c_int(n), _fp(_f32(root_lo, 3)),
                                           _fp(_f32(root_hi, 3)), C.c_void_p(stream)), "pb_upload_segments_d")

    def save_segments(self, path):
        _check(self.L.pb_save_segments(self.h, os.fsencode(path)), "pb_save_segments")

    def load_segments(self, path):
        _check(self.L.pb_load_segments(self.h, os.fsencode(path)), "pb_load_segments")

    def segment_count(self):
        n = C.c_int(0)
        _check(self.L.pb_segment_count(self.h, C.byref(n)), "pb_segment_count")
        return n.value

    def download_segments(self):
        n = self.segment_count()
        out = np.zeros(n, SEGMENT_DTYPE)
        got = C.c_int(0)
        _check(self.L.pb_download_segments(self.h, out.ctypes.data_as(C.c_void_p), C.c_int(n), C.byref(got)),
               "pb_download_segments")
        return out[:got.value]

    def root_box(self):
        lo = np.zeros(3, np.float32); hi = np.zeros(3, np.float32)
        _check(self.L.pb_get_root_box(self.h, _fp(lo), _fp(hi)), "pb_get_root_box")
        return lo, hi

    # -- photon pass -----------------------------------------------------------
    def cast_photon_beams(self, n_photons=None, seed=None):
        """CastPhotonBeams + BoundPhotonBeams (CastPhotons.h:312-390, :144-288) on the GPU."""
        n = self.control.photon_number if n_photons is None else n_photons
        seed = self.control.seed if seed is None else seed
        nb = C.c_int(0); ns = C.c_int(0)
        _check(self.L.pb_emit_beams(self.h, C.c_int(n), C.c_uint64(seed), C.byref(nb), C.byref(ns)), "pb_emit_beams")
        return nb.value, ns.value

    def recast_photon_beams(self, seed=None):
        """RecastPhotonBeams + BoundPhotonBeams: (scattered beams, segments) after re-casting the first list."""
        a = C.c_int(0); b = C.c_int(0)
        seed = self.control.seed if seed is None else seed
        _check(self.L.pb_recast_beams(self.h, C.c_uint64(seed), C.byref(a), C.byref(b)), "pb_recast_beams")
        return a.value, b.value

    def beam_counts(self):
        a = C.c_int(0); b = C.c_int(0)
        _check(self.L.pb_beam_counts(self.h, C.byref(a), C.byref(b)), "pb_beam_counts")
        return a.value, b.value

    def download_beams(self, capacity=None):
        if capacity is None:
            capacity = max(sum(self.beam_counts()), 1)
        out = np.zeros(capacity, BEAM_DTYPE)
        got = C.c_int(0)
        _check(self.L.pb_download_beams(self.h, out.ctypes.data_as(C.c_void_p), C.c_int(capacity), C.byref(got)),
               "pb_download_beams")
        return out[:got.value]

    # -- LBVH ------------------------------------------------------------------
    def build_tree(self):
        """BuildTree (CastPhotons.h:800-1126) replacement: GPU LBVH over the segment set."""
        _check(self.L.pb_build_lbvh(self.h), "pb_build_lbvh")

    def download_lbvh(self):
        n = self.segment_count()
        keys = np.zeros(n, np.uint64); perm = np.zeros(n, np.uint32); nodes = np.zeros(max(n - 1, 0), NODE_DTYPE)
        _check(self.L.pb_download_lbvh(self.h, keys.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p),
                                       nodes.ctypes.data_as(C.c_void_p)), "pb_download_lbvh")
        return keys, perm, nodes

    # -- frames ----------------------------------------------------------------
    @staticmethod
    def _cull(use_lbvh, cull):
        if cull is None:
            return 1 if use_lbvh else 0
        return {"scan": 0, "lbvh": 1, "bins": 2}[cull]

    def draw(self, x0=0, y0=0, w=None, h=None, mode=MODE_721, use_lbvh=False, cull=None):
        """DrawStandard / DrawHet (skeleton.cpp:330-412) for a pixel rectangle: (h,w,3) float32 = current/SSAA.
        cull: "scan" (every segment per tile), "lbvh" (after build_tree) or "bins" (screen-space bins)."""
        w = self.control.screen_width - x0 if w is None else w
        h = self.control.screen_height - y0 if h is None else h
        out = np.zeros((h, w, 3), np.float32)
        _check(self.L.pb_render(self.h, C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), C.c_int(mode),
                                C.c_int(self._cull(use_lbvh, cull)), _fp(out)), "pb_render")
        return out

    def draw_device(self, d_out_ptr, x0, y0, w, h, mode=MODE_721, use_lbvh=False, stream=0, cull=None):
        _check(self.L.pb_render_d(self.h, C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), C.c_int(mode),
                                  C.c_int(self._cull(use_lbvh, cull)), C.c_void_p(d_out_ptr), C.c_void_p(stream)),
               "pb_render_d")

    def draw_noalloc_device(self, d_out_ptr, x0, y0, w, h, mode=MODE_721, stream=0):
        """pb_render_noalloc_d: draw_device(cull="bins") without any allocation (fixed static workspace)."""
        _check(self.L.pb_render_noalloc_d(self.h, C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), C.c_int(mode),
                                          C.c_void_p(d_out_ptr), C.c_void_p(stream)), "pb_render_noalloc_d")

    def draw_strips_device(self, d_out_ptr, first_strip, strip_stride, n_strips, mode=MODE_721, use_lbvh=False,
                           stream=0, cull=None):
        _check(self.L.pb_render_strips_d(self.h, C.c_int(first_strip), C.c_int(strip_stride), C.c_int(n_strips),
                                         C.c_int(mode), C.c_int(self._cull(use_lbvh, cull)), C.c_void_p(d_out_ptr),
                                         C.c_void_p(stream)), "pb_render_strips_d")

    def set_precise(self, on=True):
        """Evaluate HitCylinder/Integral_721 in the reference's exact operation order (slower)."""
        _check(self.L.pb_set_precise(self.h, C.c_int(1 if on else 0)), "pb_set_precise")

    # -- counters --------------------------------------------------------------
    def enable_counters(self, on=True):
        _check(self.L.pb_enable_counters(self.h, C.c_int(1 if on else 0)), "pb_enable_counters")

    def counters(self):
        s = Stats()
        _check(self.L.pb_counters(self.h, C.byref(s)), "pb_counters")
        return s.as_dict()
