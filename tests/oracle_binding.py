"""ctypes bindings for the CHECKER libraries under oracle/ (test infrastructure only).

oracle/libptmi_oracle.so  - plain-C CPU restatement (ptmi_oracle.c)
oracle/_ref/libptmi_ref.so - the reference's own geometry headers compiled from
                             /root/reference (present only where it was built)
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("PTMI_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libptmi_oracle.so")   # override: sanitizer builds of the checker
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libptmi_ref.so")
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
                ("vfov_deg", C.c_float), ("yaw_deg", C.c_float), ("pitch_deg", C.c_float),
                ("orbit", C.c_int)]


def default_camera():
    """AppConfig defaults (application_state.h:285-286) + Sensor yaw/pitch (sensor.h:24-25)."""
    return Camera((0.5, 3.0, 8.5), (0.0, 2.5, 0.0), (0.0, 1.0, 0.0), 40.0, 90.0, 0.0, 1)


class CameraFrame(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3)]

    def as_array(self):
        return np.array(list(self.origin) + list(self.lower_left_corner) +
                        list(self.horizontal) + list(self.vertical), dtype=np.float32)


class Stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("samples", C.c_uint64), ("rays", C.c_uint64),
                ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("hits", C.c_uint64)]


class RadiosityParams(C.Structure):
    """RadiosityState / AppConfig defaults (application_state.h:207-209, 290-291)"""
    _fields_ = [("num_iterations", C.c_int), ("mc_samples", C.c_int), ("use_monte_carlo", C.c_int),
                ("enable_filtering", C.c_int), ("use_bilateral", C.c_int),
                ("filter_sigma_spatial", C.c_float), ("filter_sigma_range", C.c_float)]

    def __init__(self, num_iterations=10, mc_samples=64, use_monte_carlo=True, enable_filtering=False, use_bilateral=True,
                 filter_sigma_spatial=1.5, filter_sigma_range=0.3):
        super().__init__(num_iterations, mc_samples, int(use_monte_carlo), int(enable_filtering), int(use_bilateral),
                         filter_sigma_spatial, filter_sigma_range)


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int), ("prim", C.c_int), ("t", C.c_float), ("p", C.c_float * 3),
                ("n", C.c_float * 3), ("bsdf", C.c_float * 3), ("Le", C.c_float * 3),
                ("node_visits", C.c_int), ("prim_tests", C.c_int)]


class RefHit(C.Structure):
    _fields_ = [("hit", C.c_int), ("prim", C.c_int), ("t", C.c_float), ("p", C.c_float * 3),
                ("n", C.c_float * 3), ("bsdf", C.c_float * 3), ("Le", C.c_float * 3)]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(ORACLE_SO)
        L.po_scene_load.restype = C.c_void_p
        L.po_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.po_scene_from_arrays.restype = C.c_void_p
        L.po_scene_from_arrays.argtypes = [C.c_int] + [C.c_void_p] * 5
        L.po_scene_free.argtypes = [C.c_void_p]
        L.po_scene_num_prims.argtypes = [C.c_void_p]
        L.po_scene_num_nodes.argtypes = [C.c_void_p]
        L.po_scene_get_prims.argtypes = [C.c_void_p] * 6
        L.po_scene_get_bvh.argtypes = [C.c_void_p] * 7
        L.po_camera_frame_setup.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.POINTER(CameraFrame)]
        L.po_camera_ray.argtypes = [C.POINTER(CameraFrame), C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.po_rng_init.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        L.po_rng_uniform.restype = C.c_float
        L.po_rng_uniform.argtypes = [C.c_void_p]
        L.po_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.po_powf.restype = C.c_float
        L.po_powf.argtypes = [C.c_float, C.c_float]
        L.po_sample_cosine_hemisphere.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.po_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.POINTER(Hit)]
        L.po_render.restype = C.c_int
        L.po_scene_set_radiosity_grids.argtypes = [C.c_void_p, C.c_void_p]
        L.po_scene_set_mis_fraction.argtypes = [C.c_void_p, C.c_float]
        L.po_scene_get_cdfs.argtypes = [C.c_void_p, C.c_void_p]
        L.po_scene_set_radiosity.argtypes = [C.c_void_p, C.c_void_p]
        L.po_render_radiosity.restype = C.c_int
        L.po_render_radiosity.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.po_acosf.restype = C.c_float; L.po_acosf.argtypes = [C.c_float]
        L.po_expf.restype = C.c_float; L.po_expf.argtypes = [C.c_float]
        L.po_atan2f.restype = C.c_float; L.po_atan2f.argtypes = [C.c_float, C.c_float]
        L.po_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.po_radiosity_solve.restype = C.c_int
        L.po_radiosity_solve.argtypes = [C.c_void_p, C.POINTER(RadiosityParams), C.c_int] + [C.c_void_p] * 6
        L.po_scene_apply_grid_filter.restype = C.c_int
        L.po_scene_apply_grid_filter.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.po_form_factor_rows.restype = C.c_int
        L.po_form_factor_rows.argtypes = [C.c_void_p, C.POINTER(RadiosityParams), C.c_int, C.c_int] + [C.c_void_p] * 3
        L.po_prim_geometry.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]
        L.po_prim_sample_uniform.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.po_direction_to_grid_index.restype = C.c_int
        L.po_direction_to_grid_index.argtypes = [C.c_void_p, C.c_void_p]
        L.po_visibility_blocked.restype = C.c_int
        L.po_visibility_blocked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int]
        L.po_cdf_layout.argtypes = [C.c_void_p]; L.po_cdf_layout.restype = None
        L.po_grid_constants.argtypes = [C.c_void_p]; L.po_grid_constants.restype = None
        _oracle = L
    return _oracle


class OracleScene:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: scene load failed")
        self.h = handle
        self.L = oracle_lib()

    @classmethod
    def load(cls, path, subdivision=0, convert_quads=False):
        return cls(oracle_lib().po_scene_load(path.encode(), int(subdivision), int(bool(convert_quads))))

    @classmethod
    def from_arrays(cls, types, verts, normal, bsdf, Le):
        types = np.ascontiguousarray(types, np.int32)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4, 3)
        normal, bsdf, Le = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (normal, bsdf, Le))
        n = len(types)
        assert verts.shape[0] == n and normal.shape[0] == n
        return cls(oracle_lib().po_scene_from_arrays(n, types.ctypes.data, verts.ctypes.data, normal.ctypes.data,
                                                     bsdf.ctypes.data, Le.ctypes.data))

    def __del__(self):
        try:
            self.L.po_scene_free(self.h)
        except Exception:
            pass

    @property
    def n_prims(self):
        return self.L.po_scene_num_prims(self.h)

    @property
    def n_nodes(self):
        return self.L.po_scene_num_nodes(self.h)

    def prims(self):
        n = self.n_prims
        t = np.zeros(n, np.int32); v = np.zeros((n, 4, 3), np.float32)
        nr = np.zeros((n, 3), np.float32); b = np.zeros((n, 3), np.float32); le = np.zeros((n, 3), np.float32)
        self.L.po_scene_get_prims(self.h, t.ctypes.data, v.ctypes.data, nr.ctypes.data, b.ctypes.data, le.ctypes.data)
        return dict(type=t, verts=v, normal=nr, bsdf=b, Le=le)

    def bvh(self):
        n, m = self.n_nodes, self.n_prims
        bmin = np.zeros((n, 3), np.float32); bmax = np.zeros((n, 3), np.float32)
        left = np.zeros(n, np.int32); right = np.zeros(n, np.int32); count = np.zeros(n, np.int32)
        idx = np.zeros(m, np.int32)
        self.L.po_scene_get_bvh(self.h, bmin.ctypes.data, bmax.ctypes.data, left.ctypes.data, right.ctypes.data,
                                count.ctypes.data, idx.ctypes.data)
        return dict(bmin=bmin, bmax=bmax, left=left, right=right, count=count, indices=idx)

    def intersect(self, o, d, t_min=1e-4, t_max=3.4028234663852886e38, use_bvh=True):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        h = Hit()
        self.L.po_intersect(self.h, o.ctypes.data, d.ctypes.data, t_min, t_max, int(use_bvh), C.byref(h))
        return h

    def set_radiosity_grids(self, rgb):
        """rgb: (n_prims, 256, 3) float32 or None"""
        if rgb is None:
            self.L.po_scene_set_radiosity_grids(self.h, None); return
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.shape == (self.n_prims, 256, 3)
        self.L.po_scene_set_radiosity_grids(self.h, rgb.ctypes.data)

    def set_radiosity(self, rgb):
        if rgb is None:
            self.L.po_scene_set_radiosity(self.h, None); return
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.shape == (self.n_prims, 3)
        self.L.po_scene_set_radiosity(self.h, rgb.ctypes.data)

    def render_radiosity(self, cam, width, height, spp, seed_base=2023, y0=0, y1=None, n_threads=0, rng_state=None, reset_rng=True):
        y1 = height if y1 is None else y1
        rgb = np.zeros((height, width, 3), np.uint8); rad = np.zeros((height, width, 3), np.float32)
        rc = self.L.po_render_radiosity(self.h, C.byref(cam), width, height, spp, seed_base, int(reset_rng),
                                        None if rng_state is None else rng_state.ctypes.data, y0, y1, n_threads,
                                        rgb.ctypes.data, rad.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"po_render_radiosity failed: {rc}")
        return rgb, rad

    def radiosity_solve(self, n_threads=0, **params):
        """runSolver + precomputeCDFs (ui_windows.h:185-192); returns dict of the solver's outputs (load order)."""
        n = self.n_prims
        prm = RadiosityParams(**params)
        out = dict(form_factors=np.zeros((n, n), np.float32), radiosity=np.zeros((n, 3), np.float32),
                   unshot=np.zeros((n, 3), np.float32), grid=np.zeros((n, 256), np.float32),
                   radiosity_grid=np.zeros((n, 256, 3), np.float32))
        rays = C.c_uint64(0)
        rc = self.L.po_radiosity_solve(self.h, C.byref(prm), n_threads, out["form_factors"].ctypes.data, out["radiosity"].ctypes.data,
                                       out["unshot"].ctypes.data, out["grid"].ctypes.data, out["radiosity_grid"].ctypes.data,
                                       C.addressof(rays))
        if rc != 0:
            raise RuntimeError(f"po_radiosity_solve failed: {rc}")
        out["rays"] = rays.value
        return out

    def apply_grid_filter(self, use_bilateral=True, sigma_spatial=1.5, sigma_range=0.3):
        """"Apply Filter & Rebuild CDFs" (ui_windows.h:154-167); returns the filtered, normalised (formfactor, radiosity) pdfs"""
        ff = np.zeros((self.n_prims, 256), np.float32); rad = np.zeros((self.n_prims, 256), np.float32)
        rc = self.L.po_scene_apply_grid_filter(self.h, int(use_bilateral), sigma_spatial, sigma_range, ff.ctypes.data, rad.ctypes.data)
        if rc != 0:
            raise RuntimeError("apply_grid_filter: the scene has no radiosity grids")
        return ff, rad

    def form_factor_rows(self, rows, n_threads=0, **params):
        rows = np.ascontiguousarray(rows, np.int32)
        prm = RadiosityParams(**params)
        ff = np.zeros((len(rows), self.n_prims), np.float32); grid = np.zeros((len(rows), 256), np.float32)
        rc = self.L.po_form_factor_rows(self.h, C.byref(prm), n_threads, len(rows), rows.ctypes.data, ff.ctypes.data, grid.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"po_form_factor_rows failed: {rc}")
        return ff, grid

    def prim_geometry(self, i):
        a = C.c_float(0); c = np.zeros(3, np.float32)
        self.L.po_prim_geometry(self.h, i, C.byref(a), c.ctypes.data)
        return np.float32(a.value), c

    def sample_uniform(self, i, r1, r2):
        out = np.zeros(3, np.float32)
        self.L.po_prim_sample_uniform(self.h, i, float(r1), float(r2), out.ctypes.data)
        return out

    def set_mis_fraction(self, f):
        self.L.po_scene_set_mis_fraction(self.h, float(f))

    def cdfs(self):
        out = np.zeros((self.n_prims, 530), np.float32)
        return out if self.L.po_scene_get_cdfs(self.h, out.ctypes.data) else None

    def render(self, cam, width, height, spp, max_depth=5, seed_base=2023, y0=0, y1=None, n_threads=0,
               rng_state=None, reset_rng=True, sampling_mode=0):
        y1 = height if y1 is None else y1
        rgb = np.zeros((height, width, 3), np.uint8)
        rad = np.zeros((height, width, 3), np.float32)
        st = Stats()
        rc = self.L.po_render(self.h, C.byref(cam), width, height, spp, max_depth, int(sampling_mode), seed_base, int(reset_rng),
                              None if rng_state is None else rng_state.ctypes.data, y0, y1, n_threads,
                              rgb.ctypes.data, rad.ctypes.data, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"po_render failed: {rc}")
        return rgb, rad, st


def camera_frame(cam, width, height):
    cf = CameraFrame()
    oracle_lib().po_camera_frame_setup(C.byref(cam), width, height, C.byref(cf))
    return cf


def camera_ray(cf, u, v):
    o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
    oracle_lib().po_camera_ray(C.byref(cf), u, v, o.ctypes.data, d.ctypes.data)
    return o, d


def rng_stream(seed, subsequence, n):
    st = np.zeros(6, np.uint32)
    L = oracle_lib()
    L.po_rng_init(seed, subsequence, st.ctypes.data)
    return np.array([L.po_rng_uniform(st.ctypes.data) for _ in range(n)], np.float32), st


# ---------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.ref_scene_create.restype = C.c_void_p
        L.ref_scene_create.argtypes = [C.c_int] + [C.c_void_p] * 5
        L.ref_scene_free.argtypes = [C.c_void_p]
        L.ref_scene_num_nodes.argtypes = [C.c_void_p]
        L.ref_scene_get_bvh.argtypes = [C.c_void_p] * 7
        L.ref_tri_geometric_normal.argtypes = [C.c_void_p] * 4
        L.ref_quad_geometric_normal.argtypes = [C.c_void_p] * 5
        L.ref_centroid.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_unit_vector.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_area.restype = C.c_float
        L.ref_area.argtypes = [C.c_void_p, C.c_int]
        L.ref_sample_uniform.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.ref_intersect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                    C.c_int, C.c_void_p]
        L.ref_layout.argtypes = [C.c_void_p]; L.ref_layout.restype = None
        L.ref_grid_constants.argtypes = [C.c_void_p]; L.ref_grid_constants.restype = None
        _ref = L
    return _ref


class RefScene:
    """The reference's own Primitive[] + BVHBuilder + Scene, fed from arrays."""

    def __init__(self, types, verts, normal, bsdf, Le):
        self.L = ref_lib()
        types = np.ascontiguousarray(types, np.int32)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4, 3)
        normal, bsdf, Le = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (normal, bsdf, Le))
        self.n = len(types)
        self.h = self.L.ref_scene_create(self.n, types.ctypes.data, verts.ctypes.data, normal.ctypes.data,
                                         bsdf.ctypes.data, Le.ctypes.data)

    def __del__(self):
        try:
            self.L.ref_scene_free(self.h)
        except Exception:
            pass

    def bvh(self):
        n = self.L.ref_scene_num_nodes(self.h)
        bmin = np.zeros((n, 3), np.float32); bmax = np.zeros((n, 3), np.float32)
        left = np.zeros(n, np.int32); right = np.zeros(n, np.int32); count = np.zeros(n, np.int32)
        idx = np.zeros(self.n, np.int32)
        self.L.ref_scene_get_bvh(self.h, bmin.ctypes.data, bmax.ctypes.data, left.ctypes.data, right.ctypes.data,
                                 count.ctypes.data, idx.ctypes.data)
        return dict(bmin=bmin, bmax=bmax, left=left, right=right, count=count, indices=idx)

    def intersect(self, o, d, t_min=1e-4, t_max=3.4028234663852886e38, use_bvh=True):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = (RefHit * len(o))()
        self.L.ref_intersect(self.h, len(o), o.ctypes.data, d.ctypes.data, t_min, t_max, int(use_bvh), out)
        return out


def ref_camera(cam, width, height):
    out = np.zeros(12, np.float32)
    lf = np.array(list(cam.origin), np.float32); la = np.array(list(cam.lookat), np.float32)
    up = np.array(list(cam.vup), np.float32)
    ref_lib().ref_camera(lf.ctypes.data, la.ctypes.data, up.ctypes.data, cam.vfov_deg, cam.yaw_deg, cam.pitch_deg,
                         cam.orbit, width, height, out.ctypes.data)
    return out


# ---- the reference's PBRT import, compiled (oracle/_ref/libptmi_ref_pbrt.so) -------------------------------------------------
REF_PBRT_SO = os.path.join(os.path.dirname(REF_SO), "libptmi_ref_pbrt.so")
_ref_pbrt = None


def ref_pbrt_available():
    return os.path.exists(REF_PBRT_SO)


def ref_pbrt_load(path):
    """loadPBRT of the reference on `path`: dict(type, verts, normal, bsdf, Le) or None where it fails (returns false / throws)."""
    global _ref_pbrt
    if _ref_pbrt is None:
        L = C.CDLL(REF_PBRT_SO)
        L.ref_pbrt_load.restype = C.c_void_p; L.ref_pbrt_load.argtypes = [C.c_char_p, C.c_int]
        L.ref_pbrt_count.argtypes = [C.c_void_p]; L.ref_pbrt_get.argtypes = [C.c_void_p] * 6; L.ref_pbrt_free.argtypes = [C.c_void_p]
        _ref_pbrt = L
    L = _ref_pbrt
    h = L.ref_pbrt_load(os.fsencode(os.path.abspath(path)), 1)
    if not h:
        return None
    n = L.ref_pbrt_count(h)
    t = np.zeros(n, np.int32); v = np.zeros((n, 4, 3), np.float32)
    nr = np.zeros((n, 3), np.float32); b = np.zeros((n, 3), np.float32); le = np.zeros((n, 3), np.float32)
    L.ref_pbrt_get(h, t.ctypes.data, v.ctypes.data, nr.ctypes.data, b.ctypes.data, le.ctypes.data)
    L.ref_pbrt_free(h)
    return dict(type=t, verts=v, normal=nr, bsdf=b, Le=le)


# ---- the reference's OBJ/MTL loader, compiled (oracle/_ref/libptmi_ref_obj.so; needs NVIDIA's own cuda_runtime.h at build time) ----
REF_OBJ_SO = os.path.join(os.path.dirname(REF_SO), "libptmi_ref_obj.so")
_ref_obj = None


def ref_obj_available():
    return os.path.exists(REF_OBJ_SO)


def ref_obj_load(path):
    """loadOBJ (+ loadMTL) of the reference on `path`: dict(type, verts, normal, bsdf, Le, warnings) or None where it returns false."""
    global _ref_obj
    if _ref_obj is None:
        L = C.CDLL(REF_OBJ_SO)
        L.ref_obj_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.ref_obj_warning_count.argtypes = [C.c_void_p]
        L.ref_obj_get.argtypes = [C.c_void_p] * 6; L.ref_obj_free.argtypes = [C.c_void_p]; L.ref_obj_free.restype = None
        _ref_obj = L
    L = _ref_obj
    n = C.c_int(); h = C.c_void_p()
    ok = L.ref_obj_load(os.fsencode(path), C.byref(n), C.byref(h))
    out = None
    if ok:
        t = np.zeros(n.value, np.int32); v = np.zeros((n.value, 4, 3), np.float32)
        nr = np.zeros((n.value, 3), np.float32); b = np.zeros((n.value, 3), np.float32); le = np.zeros((n.value, 3), np.float32)
        if n.value:
            L.ref_obj_get(h, t.ctypes.data, v.ctypes.data, nr.ctypes.data, b.ctypes.data, le.ctypes.data)
        out = dict(type=t, verts=v, normal=nr, bsdf=b, Le=le, warnings=L.ref_obj_warning_count(h))
    L.ref_obj_free(h)
    return out
