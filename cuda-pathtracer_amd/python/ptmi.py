"""ctypes binding of libptmi.so (include/ptmi.h) — the product path.

No fallback: if the shared library is missing or no gfx950 device is usable, construction
raises.  Mirrors the reference's three host entry points:

    Renderer.load_scene(path, subdivision, convert_quads)   ~ SceneState::loadScene
    Renderer.update_resolution(w, h, tiling)                 ~ RenderState::updateResolution
    Renderer.render_frame()                                  ~ renderFrame()
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTMI_LIB") or os.path.join(os.path.dirname(_HERE), "libptmi.so")   # PTMI_LIB: A/B experiments only

EXPORTS = [
    "ptmi_ctx_create", "ptmi_ctx_destroy", "ptmi_last_error", "ptmi_default_camera", "ptmi_default_config",
    "ptmi_default_tiling", "ptmi_load_scene", "ptmi_load_scene_arrays", "ptmi_scene_info", "ptmi_scene_get_prims",
    "ptmi_scene_get_bvh", "ptmi_update_resolution", "ptmi_set_camera", "ptmi_set_config", "ptmi_get_camera_frame",
    "ptmi_local_rows", "ptmi_local_row_map", "ptmi_render_frame", "ptmi_device_image", "ptmi_read_image",
    "ptmi_copy_image_device", "ptmi_set_radiosity_grids", "ptmi_get_precomputed_cdfs", "ptmi_set_radiosity",
           "ptmi_write_png", "ptmi_apply_grid_filter", "ptmi_use_raw_cdfs", "ptmi_get_filtered_pdfs", "ptmi_default_radiosity_params", "ptmi_run_radiosity_solver", "ptmi_get_radiosity_solution",
    "ptmi_debug_intersect", "ptmi_debug_rng", "ptmi_debug_cosine_sample", "ptmi_debug_set_traversal", "ptmi_debug_rcp_check",
    "ptmi_host_scene_load", "ptmi_host_scene_from_arrays", "ptmi_host_scene_free", "ptmi_host_scene_info",
    "ptmi_host_scene_get_prims", "ptmi_host_scene_get_bvh", "ptmi_host_camera_frame", "ptmi_host_local_row_map",
    "ptmi_host_cdf_record_layout", "ptmi_host_image",
    "ptmi_dist_unique_id", "ptmi_dist_init", "ptmi_dist_finalize", "ptmi_gather_frame", "ptmi_gather_wait", "ptmi_frame_device",
    "ptmi_read_frame", "ptmi_dist_barrier", "ptmi_dist_allreduce_max", "ptmi_debug_place_tiles", "ptmi_debug_set_packed_min_nodes", "ptmi_debug_set_packed_top", "ptmi_render_frames", "ptmi_select_frame",
    "ptmi_debug_set_fast_tree", "ptmi_debug_intersect_fast", "ptmi_dist_comm_count", "ptmi_host_fast_tree_build", "ptmi_host_fast_tree_intersect", "ptmi_host_fast_tree_stats",
    "ptmi_debug_set_solver_walk", "ptmi_debug_get_traversal",
]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
                ("vfov_deg", C.c_float), ("yaw_deg", C.c_float), ("pitch_deg", C.c_float), ("orbit", C.c_int)]


class Config(C.Structure):
    _fields_ = [("spp", C.c_int), ("max_depth", C.c_int), ("sampling_mode", C.c_int), ("seed_base", C.c_uint64),
                ("segments_per_launch", C.c_int), ("collect_stats", C.c_int), ("wave_tiles", C.c_int), ("streams", C.c_int), ("mis_bsdf_fraction", C.c_float), ("integrator", C.c_int),
                ("download_image", C.c_int), ("fast_tree", C.c_int)]


class Tiling(C.Structure):
    _fields_ = [("n_ranks", C.c_int), ("rank", C.c_int), ("row_block", C.c_int)]


class RadiosityParams(C.Structure):
    _fields_ = [("num_iterations", C.c_int), ("mc_samples", C.c_int), ("use_monte_carlo", C.c_int),
                ("enable_filtering", C.c_int), ("use_bilateral", C.c_int),
                ("filter_sigma_spatial", C.c_float), ("filter_sigma_range", C.c_float)]


class RadiosityStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("form_factor_ms", C.c_double), ("iteration_ms", C.c_double), ("grid_ms", C.c_double),
                ("pairs", C.c_uint64), ("rays", C.c_uint64), ("cert_chain", C.c_uint64), ("cert_fallback", C.c_uint64), ("walk", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("bounce_kernel_ms", C.c_double), ("bounce_launches", C.c_uint64), ("path_visits", C.c_uint64),
                ("samples", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64), ("hits", C.c_uint64), ("top_node_visits", C.c_uint64),
                ("cert_chain", C.c_uint64), ("cert_fallback", C.c_uint64)]


class PtmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptmi error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Loads libptmi.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C cuda-pathtracer_amd` "
                               f"(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.ptmi_last_error.restype = C.c_char_p
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.c_void_p
        L.ptmi_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ptmi_ctx_destroy.argtypes = [vp]; L.ptmi_ctx_destroy.restype = None
        L.ptmi_default_camera.argtypes = [C.POINTER(Camera)]; L.ptmi_default_camera.restype = None
        L.ptmi_default_config.argtypes = [C.POINTER(Config)]; L.ptmi_default_config.restype = None
        L.ptmi_default_tiling.argtypes = [C.POINTER(Tiling)]; L.ptmi_default_tiling.restype = None
        L.ptmi_load_scene.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.ptmi_load_scene_arrays.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
        L.ptmi_scene_info.argtypes = [vp, ip, ip, ip, ip, ip]
        L.ptmi_scene_get_prims.argtypes = [vp] * 6
        L.ptmi_scene_get_bvh.argtypes = [vp] * 7
        L.ptmi_update_resolution.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Tiling)]
        L.ptmi_set_camera.argtypes = [vp, C.POINTER(Camera)]
        L.ptmi_set_config.argtypes = [vp, C.POINTER(Config)]
        L.ptmi_get_camera_frame.argtypes = [vp, fp]
        L.ptmi_local_rows.argtypes = [vp, ip]
        L.ptmi_local_row_map.argtypes = [vp, vp]
        L.ptmi_render_frame.argtypes = [vp, C.POINTER(Stats)]
        L.ptmi_render_frames.argtypes = [vp, C.c_int, C.POINTER(Stats)]
        L.ptmi_select_frame.argtypes = [vp, C.c_int]
        L.ptmi_device_image.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.ptmi_read_image.argtypes = [vp, vp, vp]
        L.ptmi_copy_image_device.argtypes = [vp, vp, vp]
        L.ptmi_set_radiosity_grids.argtypes = [vp, C.c_int, vp]
        L.ptmi_get_precomputed_cdfs.argtypes = [vp, vp]
        L.ptmi_set_radiosity.argtypes = [vp, C.c_int, vp]
        L.ptmi_apply_grid_filter.argtypes = [vp, C.c_int, C.c_float, C.c_float]
        L.ptmi_use_raw_cdfs.argtypes = [vp]
        L.ptmi_get_filtered_pdfs.argtypes = [vp, vp, vp]
        L.ptmi_default_radiosity_params.argtypes = [C.POINTER(RadiosityParams)]
        L.ptmi_default_radiosity_params.restype = None
        L.ptmi_run_radiosity_solver.argtypes = [vp, C.POINTER(RadiosityParams), C.POINTER(RadiosityStats)]
        L.ptmi_get_radiosity_solution.argtypes = [vp] * 6
        L.ptmi_debug_intersect.argtypes = [vp, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp]
        L.ptmi_debug_rng.argtypes = [vp, C.c_uint64, C.c_int, vp, C.c_int, vp]
        L.ptmi_debug_cosine_sample.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.ptmi_debug_set_traversal.argtypes = [vp, C.c_int, C.c_int, ip]
        L.ptmi_debug_set_solver_walk.argtypes = [vp, C.c_int, C.c_int]
        L.ptmi_debug_get_traversal.argtypes = [vp, ip]
        L.ptmi_debug_rcp_check.argtypes = [vp, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.ptmi_host_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        L.ptmi_host_scene_from_arrays.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.POINTER(vp)]
        L.ptmi_host_scene_free.argtypes = [vp]; L.ptmi_host_scene_free.restype = None
        L.ptmi_host_scene_info.argtypes = [vp, ip, ip, ip, ip, ip]
        L.ptmi_host_scene_get_prims.argtypes = [vp] * 6
        L.ptmi_host_scene_get_bvh.argtypes = [vp] * 7
        L.ptmi_host_camera_frame.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, vp]
        L.ptmi_host_local_row_map.argtypes = [C.c_int, C.POINTER(Tiling), ip, vp]
        L.ptmi_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, vp]
        L.ptmi_host_cdf_record_layout.argtypes = [vp]
        L.ptmi_host_image.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.ptmi_dist_unique_id.argtypes = [vp]
        L.ptmi_dist_init.argtypes = [vp, vp, C.c_int, C.c_int]
        L.ptmi_dist_finalize.argtypes = [vp]
        L.ptmi_gather_frame.argtypes = [vp, C.c_int, C.c_int]
        L.ptmi_gather_wait.argtypes = [vp]
        L.ptmi_frame_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.ptmi_read_frame.argtypes = [vp, vp, vp]
        L.ptmi_dist_barrier.argtypes = [vp]
        L.ptmi_dist_comm_count.argtypes = [vp, ip]
        L.ptmi_dist_allreduce_max.argtypes = [vp, C.POINTER(C.c_double)]
        L.ptmi_debug_set_packed_min_nodes.argtypes = [vp, C.c_int, ip]
        L.ptmi_debug_set_packed_top.argtypes = [vp, C.c_int, ip, ip]
        L.ptmi_debug_place_tiles.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
        L.ptmi_debug_set_fast_tree.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_int, ip, ip, ip]
        L.ptmi_debug_intersect_fast.argtypes = [vp, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp]
        L.ptmi_host_fast_tree_build.argtypes = [vp, C.c_int, C.c_float, C.c_float, ip, ip, C.POINTER(C.c_double)]
        L.ptmi_host_fast_tree_stats.argtypes = [vp, vp]
        L.ptmi_host_fast_tree_intersect.argtypes = [vp, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise PtmiError(rc, lib().ptmi_last_error().decode(errors="replace"))


class HostScene:
    """Host half of loadScene (parse, convert, subdivide, BVH) - needs no GPU."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def load(cls, filename, subdivision_count=0, convert_quads=False):
        h = C.c_void_p()
        _check(lib().ptmi_host_scene_load(os.fsencode(filename), int(subdivision_count), int(bool(convert_quads)), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, types, verts, normal, bsdf, Le):
        types = np.ascontiguousarray(types, np.int32)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4, 3)
        normal, bsdf, Le = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (normal, bsdf, Le))
        h = C.c_void_p()
        _check(lib().ptmi_host_scene_from_arrays(len(types), types.ctypes.data, verts.ctypes.data, normal.ctypes.data,
                                                 bsdf.ctypes.data, Le.ctypes.data, C.byref(h)))
        return cls(h)

    def __del__(self):
        try:
            if self.h and self.h.value:
                lib().ptmi_host_scene_free(self.h)
        except Exception:
            pass

    def info(self):
        v = [C.c_int() for _ in range(5)]
        _check(lib().ptmi_host_scene_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("n_prims", "n_tris", "n_quads", "n_bvh_nodes", "bvh_depth"), (x.value for x in v)))

    def prims(self):
        n = self.info()["n_prims"]
        t = np.zeros(n, np.int32); v = np.zeros((n, 4, 3), np.float32)
        nr = np.zeros((n, 3), np.float32); b = np.zeros((n, 3), np.float32); le = np.zeros((n, 3), np.float32)
        _check(lib().ptmi_host_scene_get_prims(self.h, t.ctypes.data, v.ctypes.data, nr.ctypes.data, b.ctypes.data, le.ctypes.data))
        return dict(type=t, verts=v, normal=nr, bsdf=b, Le=le)

    def bvh(self):
        info = self.info(); n, m = info["n_bvh_nodes"], info["n_prims"]
        bmin = np.zeros((n, 3), np.float32); bmax = np.zeros((n, 3), np.float32)
        left = np.zeros(n, np.int32); right = np.zeros(n, np.int32); count = np.zeros(n, np.int32); idx = np.zeros(m, np.int32)
        _check(lib().ptmi_host_scene_get_bvh(self.h, bmin.ctypes.data, bmax.ctypes.data, left.ctypes.data, right.ctypes.data,
                                             count.ctypes.data, idx.ctypes.data))
        return dict(bmin=bmin, bmax=bmax, left=left, right=right, count=count, indices=idx)


    # --- the opt-in fast tree (csrc/wide_bvh.h), host halves ---
    def fast_tree_build(self, max_leaf=3, c_trav=1.0, c_tri=1.0):
        n = C.c_int(); d = C.c_int(); sah = C.c_double()
        _check(lib().ptmi_host_fast_tree_build(self.h, int(max_leaf), float(c_trav), float(c_tri), C.byref(n), C.byref(d), C.byref(sah)))
        return dict(n_nodes=n.value, depth=d.value, sah=sah.value)

    def fast_tree_stats(self):
        out = np.zeros(13, np.int32)
        _check(lib().ptmi_host_fast_tree_stats(self.h, out.ctypes.data))
        return dict(nodes_by_children=out[:9].tolist(), leaves_by_triangles=out[9:].tolist())

    def fast_tree_intersect(self, o, d, t_min=1e-4, t_max=3.4028234663852886e38):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        prim = np.zeros(n, np.int32); t = np.zeros(n, np.float32); counts = np.zeros(3, np.uint64)
        _check(lib().ptmi_host_fast_tree_intersect(self.h, n, o.ctypes.data, d.ctypes.data, t_min, t_max, prim.ctypes.data, t.ctypes.data,
                                                   counts.ctypes.data))
        return dict(prim=prim, t=t, node_visits=int(counts[0]), prim_tests=int(counts[1]), max_stack=int(counts[2]))


def host_camera_frame(cam, width, height):
    out = np.zeros(12, np.float32)
    _check(lib().ptmi_host_camera_frame(C.byref(cam), int(width), int(height), out.ctypes.data))
    return out


def host_local_row_map(height, n_ranks, rank, row_block):
    t = Tiling(int(n_ranks), int(rank), int(row_block)); n = C.c_int()
    _check(lib().ptmi_host_local_row_map(int(height), C.byref(t), C.byref(n), None))
    rows = np.zeros(n.value, np.int32)
    if n.value:
        _check(lib().ptmi_host_local_row_map(int(height), C.byref(t), C.byref(n), rows.ctypes.data))
    return rows


def host_cdf_record_layout():
    out = np.zeros(10, np.int32)
    _check(lib().ptmi_host_cdf_record_layout(out.ctypes.data))
    return out


def default_camera():
    c = Camera(); lib().ptmi_default_camera(C.byref(c)); return c


def default_config():
    c = Config(); lib().ptmi_default_config(C.byref(c)); return c


class Renderer:
    """One ApplicationState bound to one GPU."""

    def __init__(self, device_id=0):
        self.L = lib()
        self.h = C.c_void_p()
        self._ck(self.L.ptmi_ctx_create(int(device_id), C.byref(self.h)))
        self.width = self.height = 0
        self.config = default_config()

    def _ck(self, rc):
        if rc != 0:
            raise PtmiError(rc, self.L.ptmi_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.ptmi_ctx_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- SceneState::loadScene ---
    def load_scene(self, filename, subdivision_count=0, convert_quads=False):
        self._ck(self.L.ptmi_load_scene(self.h, os.fsencode(filename), int(subdivision_count), int(bool(convert_quads))))

    def load_scene_arrays(self, types, verts, normal, bsdf, Le):
        types = np.ascontiguousarray(types, np.int32)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4, 3)
        normal, bsdf, Le = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (normal, bsdf, Le))
        assert len(verts) == len(types) == len(normal) == len(bsdf) == len(Le)
        self._ck(self.L.ptmi_load_scene_arrays(self.h, len(types), types.ctypes.data, verts.ctypes.data,
                                               normal.ctypes.data, bsdf.ctypes.data, Le.ctypes.data))

    def set_radiosity_grids(self, rgb):
        """rgb: (n_prims, 256, 3) float32 radiosity grids in load order, or None to drop the CDF records."""
        if rgb is None:
            self._ck(self.L.ptmi_set_radiosity_grids(self.h, 0, None)); return
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.ndim == 3 and rgb.shape[1:] == (256, 3)
        self._ck(self.L.ptmi_set_radiosity_grids(self.h, rgb.shape[0], rgb.ctypes.data))

    def set_radiosity(self, rgb):
        """rgb: (n_prims, 3) float32 per-primitive radiosity in load order, or None (zero)."""
        if rgb is None:
            self._ck(self.L.ptmi_set_radiosity(self.h, 0, None)); return
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.ndim == 2 and rgb.shape[1] == 3
        self._ck(self.L.ptmi_set_radiosity(self.h, rgb.shape[0], rgb.ctypes.data))

    def apply_grid_filter(self, use_bilateral=True, sigma_spatial=1.5, sigma_range=0.3):
        """"Apply Filter & Rebuild CDFs" (ui_windows.h:154-167); returns the (formfactor, radiosity) filtered pdfs"""
        self._ck(self.L.ptmi_apply_grid_filter(self.h, int(use_bilateral), sigma_spatial, sigma_range))
        n = self.scene_info()["n_prims"]
        ff = np.zeros((n, 256), np.float32); rad = np.zeros((n, 256), np.float32)
        self._ck(self.L.ptmi_get_filtered_pdfs(self.h, ff.ctypes.data, rad.ctypes.data))
        return ff, rad

    def use_raw_cdfs(self):
        self._ck(self.L.ptmi_use_raw_cdfs(self.h))

    def run_radiosity_solver(self, **params):
        """RadiosityState::runSolver + precomputeCDFs + primitive upload (ui_windows.h:185-192).  Keyword arguments are
        the fields of ptmi_radiosity_params; defaults are the reference's."""
        prm = RadiosityParams()
        self.L.ptmi_default_radiosity_params(C.byref(prm))
        for k, v in params.items():
            if k not in dict(RadiosityParams._fields_):
                raise TypeError(f"unknown radiosity parameter {k}")
            setattr(prm, k, type(getattr(prm, k))(v))
        st = RadiosityStats()
        self._ck(self.L.ptmi_run_radiosity_solver(self.h, C.byref(prm), C.byref(st)))
        return st

    def radiosity_solution(self, form_factors=True):
        n = self.scene_info()["n_prims"]
        out = dict(radiosity=np.zeros((n, 3), np.float32), unshot=np.zeros((n, 3), np.float32),
                   grid=np.zeros((n, 256), np.float32), radiosity_grid=np.zeros((n, 256, 3), np.float32))
        if form_factors:
            out["form_factors"] = np.zeros((n, n), np.float32)
        self._ck(self.L.ptmi_get_radiosity_solution(self.h, out["form_factors"].ctypes.data if form_factors else None,
                                                    out["radiosity"].ctypes.data, out["unshot"].ctypes.data,
                                                    out["grid"].ctypes.data, out["radiosity_grid"].ctypes.data))
        return out

    def precomputed_cdfs(self):
        out = np.zeros((self.scene_info()["n_prims"], 530), np.float32)
        self._ck(self.L.ptmi_get_precomputed_cdfs(self.h, out.ctypes.data))
        return out

    def scene_info(self):
        v = [C.c_int() for _ in range(5)]
        self._ck(self.L.ptmi_scene_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("n_prims", "n_tris", "n_quads", "n_bvh_nodes", "bvh_depth"), (x.value for x in v)))

    def scene_prims(self):
        n = self.scene_info()["n_prims"]
        t = np.zeros(n, np.int32); v = np.zeros((n, 4, 3), np.float32)
        nr = np.zeros((n, 3), np.float32); b = np.zeros((n, 3), np.float32); le = np.zeros((n, 3), np.float32)
        self._ck(self.L.ptmi_scene_get_prims(self.h, t.ctypes.data, v.ctypes.data, nr.ctypes.data, b.ctypes.data, le.ctypes.data))
        return dict(type=t, verts=v, normal=nr, bsdf=b, Le=le)

    def scene_bvh(self):
        info = self.scene_info(); n, m = info["n_bvh_nodes"], info["n_prims"]
        bmin = np.zeros((n, 3), np.float32); bmax = np.zeros((n, 3), np.float32)
        left = np.zeros(n, np.int32); right = np.zeros(n, np.int32); count = np.zeros(n, np.int32); idx = np.zeros(m, np.int32)
        self._ck(self.L.ptmi_scene_get_bvh(self.h, bmin.ctypes.data, bmax.ctypes.data, left.ctypes.data, right.ctypes.data,
                                           count.ctypes.data, idx.ctypes.data))
        return dict(bmin=bmin, bmax=bmax, left=left, right=right, count=count, indices=idx)

    # --- RenderState::updateResolution / allocateBuffers ---
    def update_resolution(self, width, height, n_ranks=1, rank=0, row_block=8):
        t = Tiling(int(n_ranks), int(rank), int(row_block))
        self._ck(self.L.ptmi_update_resolution(self.h, int(width), int(height), C.byref(t)))
        self.width, self.height = int(width), int(height)

    def set_camera(self, cam):
        self._ck(self.L.ptmi_set_camera(self.h, C.byref(cam)))

    def set_config(self, spp=None, max_depth=None, seed_base=None, segments_per_launch=None, collect_stats=None, wave_tiles=None, streams=None,
                   sampling_mode=None, mis_bsdf_fraction=None, integrator=None, download_image=None, fast_tree=None):
        c = self.config
        if spp is not None: c.spp = int(spp)
        if max_depth is not None: c.max_depth = int(max_depth)
        if seed_base is not None: c.seed_base = int(seed_base)
        if segments_per_launch is not None: c.segments_per_launch = int(segments_per_launch)
        if collect_stats is not None: c.collect_stats = int(bool(collect_stats))
        if wave_tiles is not None: c.wave_tiles = int(bool(wave_tiles))
        if streams is not None: c.streams = int(streams)
        if sampling_mode is not None: c.sampling_mode = int(sampling_mode)
        if mis_bsdf_fraction is not None: c.mis_bsdf_fraction = float(mis_bsdf_fraction)
        if integrator is not None: c.integrator = int(integrator)
        if download_image is not None: c.download_image = int(bool(download_image))
        if fast_tree is not None: c.fast_tree = int(bool(fast_tree))
        self._ck(self.L.ptmi_set_config(self.h, C.byref(c)))

    def camera_frame(self):
        out = np.zeros(12, np.float32)
        self._ck(self.L.ptmi_get_camera_frame(self.h, out.ctypes.data)); return out

    def local_rows(self):
        n = C.c_int(); self._ck(self.L.ptmi_local_rows(self.h, C.byref(n)))
        rows = np.zeros(n.value, np.int32)
        if n.value: self._ck(self.L.ptmi_local_row_map(self.h, rows.ctypes.data))
        return rows

    # --- renderFrame ---
    def render_frame(self, want_stats=True):
        st = Stats()
        self._ck(self.L.ptmi_render_frame(self.h, C.byref(st) if want_stats else None))
        return st

    def render_frames(self, n_frames, want_stats=True):
        """n_frames successive frames as one pipelined run (ptmi_render_frames); the image buffers then hold the last one."""
        st = Stats()
        self._ck(self.L.ptmi_render_frames(self.h, int(n_frames), C.byref(st) if want_stats else None))
        return st

    def select_frame(self, frame):
        self._ck(self.L.ptmi_select_frame(self.h, int(frame)))

    def device_image(self):
        a = C.c_void_p(); b = C.c_void_p()
        self._ck(self.L.ptmi_device_image(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def read_image(self, rgb8=True, radiance=True):
        n = len(self.local_rows())
        rgb = np.zeros((n, self.width, 3), np.uint8) if rgb8 else None
        rad = np.zeros((n, self.width, 3), np.float32) if radiance else None
        self._ck(self.L.ptmi_read_image(self.h, rgb.ctypes.data if rgb8 else None, rad.ctypes.data if radiance else None))
        return rgb, rad

    def copy_image_device(self, d_rgb8_ptr=None, d_radiance_ptr=None):
        """D2D copy of the local rows into caller-owned device buffers (raw pointers, e.g. tensor.data_ptr())."""
        self._ck(self.L.ptmi_copy_image_device(self.h, d_rgb8_ptr, d_radiance_ptr))

    def host_image(self):
        """RenderState::h_image: the pinned host copy of the local 8-bit rows that render_frame fills when download_image is set
        (a view, valid until the next update_resolution / close)."""
        p = C.c_void_p(); n = C.c_uint64()
        self._ck(self.L.ptmi_host_image(self.h, C.byref(p), C.byref(n)))
        buf = (C.c_ubyte * n.value).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(-1, self.width, 3)

    # --- multi-GPU frame exchange (RCCL behind the C ABI) ---
    GATHER_RGB8, GATHER_RADIANCE = 1, 2

    @staticmethod
    def dist_unique_id():
        """ncclGetUniqueId: 128 bytes to be shipped from one rank to all ranks (any channel)."""
        buf = C.create_string_buffer(128)
        _check(lib().ptmi_dist_unique_id(buf))
        return buf.raw

    def dist_init(self, unique_id, n_ranks, rank):
        assert len(unique_id) == 128
        self._ck(self.L.ptmi_dist_init(self.h, C.c_char_p(unique_id), int(n_ranks), int(rank)))

    def dist_finalize(self):
        self._ck(self.L.ptmi_dist_finalize(self.h))

    def gather_frame(self, dst_rank=0, what=3):
        """The one exchange step of a frame; only enqueues (see include/ptmi.h)."""
        self._ck(self.L.ptmi_gather_frame(self.h, int(dst_rank), int(what)))

    def gather_wait(self):
        self._ck(self.L.ptmi_gather_wait(self.h))

    def read_frame(self, rgb8=True, radiance=True):
        rgb = np.zeros((self.height, self.width, 3), np.uint8) if rgb8 else None
        rad = np.zeros((self.height, self.width, 3), np.float32) if radiance else None
        self._ck(self.L.ptmi_read_frame(self.h, rgb.ctypes.data if rgb8 else None, rad.ctypes.data if radiance else None))
        return rgb, rad

    def dist_comm_count(self):
        n = C.c_int(); self._ck(self.L.ptmi_dist_comm_count(self.h, C.byref(n))); return n.value

    def dist_barrier(self):
        self._ck(self.L.ptmi_dist_barrier(self.h))

    def dist_allreduce_max(self, value):
        v = C.c_double(float(value))
        self._ck(self.L.ptmi_dist_allreduce_max(self.h, C.byref(v)))
        return v.value

    def debug_place_tiles(self, width, height, n_ranks, row_block, tiles_rgb8=None, tiles_radiance=None):
        """dst-side row placement of the gather alone: tiles concatenated in rank order -> frame."""
        n = width * height * 3
        out_rgb = out_rad = None
        if tiles_rgb8 is not None:
            tiles_rgb8 = np.ascontiguousarray(tiles_rgb8, np.uint8).reshape(-1); assert tiles_rgb8.size == n
            out_rgb = np.zeros((height, width, 3), np.uint8)
        if tiles_radiance is not None:
            tiles_radiance = np.ascontiguousarray(tiles_radiance, np.float32).reshape(-1); assert tiles_radiance.size == n
            out_rad = np.zeros((height, width, 3), np.float32)
        self._ck(self.L.ptmi_debug_place_tiles(self.h, width, height, n_ranks, row_block,
                                               tiles_rgb8.ctypes.data if tiles_rgb8 is not None else None,
                                               tiles_radiance.ctypes.data if tiles_radiance is not None else None,
                                               out_rgb.ctypes.data if out_rgb is not None else None,
                                               out_rad.ctypes.data if out_rad is not None else None))
        return out_rgb, out_rad

    # --- test hooks ---
    SWEEP, LANE, STACK, PHASED, PACKED, WIDE, CERTIFIED = 0, 1, 2, 3, 4, 5, 6

    def set_traversal(self, force_mode=-1, sweep_max_prims=64):
        """Returns the traversal mode in effect for the loaded scene (-1 if none)."""
        m = C.c_int(-1)
        self._ck(self.L.ptmi_debug_set_traversal(self.h, int(force_mode), int(sweep_max_prims), C.byref(m)))
        return m.value

    def traversal(self):
        """The traversal mode in effect for the loaded scene (-1 if none); changes nothing."""
        m = C.c_int(-1)
        self._ck(self.L.ptmi_debug_get_traversal(self.h, C.byref(m)))
        return m.value

    def set_solver_walk(self, force_walk=-1, min_prims=256):
        """Visibility walk of the next radiosity solve: -1 automatic, 0 the reference's tree, 2 certified (identical form factors)."""
        self._ck(self.L.ptmi_debug_set_solver_walk(self.h, int(force_walk), int(min_prims)))

    def set_packed_min_nodes(self, min_nodes=8192):
        """Smallest tree (BVH nodes) that gets the packed layout of traversal mode PACKED; returns the record positions built."""
        n = C.c_int()
        self._ck(self.L.ptmi_debug_set_packed_min_nodes(self.h, int(min_nodes), C.byref(n)))
        return n.value

    def set_packed_top(self, top_records=512):
        """Record positions of the packed tree kept in LDS; returns (n_top, top_depth) built for the loaded scene."""
        a = C.c_int(); b = C.c_int()
        self._ck(self.L.ptmi_debug_set_packed_top(self.h, int(top_records), C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_rcp_check(self, first_bits, count):
        bad = C.c_uint64(); first = C.c_uint32()
        self._ck(self.L.ptmi_debug_rcp_check(self.h, int(first_bits), int(count), C.byref(bad), C.byref(first)))
        return bad.value, first.value

    def debug_intersect(self, o, d, t_min=1e-4, t_max=3.4028234663852886e38):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.int32); prim = np.zeros(n, np.int32); t = np.zeros(n, np.float32)
        p = np.zeros((n, 3), np.float32); nr = np.zeros((n, 3), np.float32)
        self._ck(self.L.ptmi_debug_intersect(self.h, n, o.ctypes.data, d.ctypes.data, t_min, t_max, hit.ctypes.data,
                                             prim.ctypes.data, t.ctypes.data, p.ctypes.data, nr.ctypes.data))
        return dict(hit=hit, prim=prim, t=t, p=p, n=nr)

    def debug_set_fast_tree(self, max_leaf=3, c_trav=1.0, c_tri=1.0, top_nodes=80):
        n = C.c_int(); d = C.c_int(); t = C.c_int()
        self._ck(self.L.ptmi_debug_set_fast_tree(self.h, int(max_leaf), float(c_trav), float(c_tri), int(top_nodes), C.byref(n), C.byref(d), C.byref(t)))
        return dict(n_nodes=n.value, depth=d.value, n_top=t.value)

    def debug_intersect_fast(self, o, d, t_min=1e-4, t_max=3.4028234663852886e38):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.int32); prim = np.zeros(n, np.int32); t = np.zeros(n, np.float32); counts = np.zeros(2, np.uint64)
        self._ck(self.L.ptmi_debug_intersect_fast(self.h, n, o.ctypes.data, d.ctypes.data, t_min, t_max, hit.ctypes.data,
                                                  prim.ctypes.data, t.ctypes.data, counts.ctypes.data))
        return dict(hit=hit, prim=prim, t=t, node_visits=int(counts[0]), prim_tests=int(counts[1]))

    def debug_rng(self, seed_base, pixels, count):
        pixels = np.ascontiguousarray(pixels, np.int32)
        out = np.zeros((len(pixels), count), np.float32)
        self._ck(self.L.ptmi_debug_rng(self.h, int(seed_base), len(pixels), pixels.ctypes.data, int(count), out.ctypes.data))
        return out

    def debug_cosine_sample(self, normals, u, v):
        normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        u = np.ascontiguousarray(u, np.float32); v = np.ascontiguousarray(v, np.float32)
        out = np.zeros_like(normals)
        self._ck(self.L.ptmi_debug_cosine_sample(self.h, len(u), normals.ctypes.data, u.ctypes.data, v.ctypes.data, out.ctypes.data))
        return out


def write_png(path, rgb8):
    """"Save PNG" (ui_windows.h:195-210): rgb8 is (H, W, 3) uint8 with row 0 = bottom row, as read_image returns it."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    assert rgb8.ndim == 3 and rgb8.shape[2] == 3
    _check(lib().ptmi_write_png(os.fsencode(path), rgb8.shape[1], rgb8.shape[0], rgb8.ctypes.data))


def render_image(scene_path, width, height, spp, max_depth=5, camera=None, device_id=0, **cfg):
    """Convenience: load + allocate + one frame; returns (rgb8, radiance, stats) of the full frame."""
    r = Renderer(device_id)
    r.load_scene(scene_path)
    if camera is not None: r.set_camera(camera)
    r.update_resolution(width, height)
    r.set_config(spp=spp, max_depth=max_depth, **cfg)
    st = r.render_frame()
    rgb, rad = r.read_image()
    r.close()
    return rgb, rad, st
