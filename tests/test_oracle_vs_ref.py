"""oracle/ptmi_oracle.c against the REFERENCE's own geometry headers compiled from
/root/reference into oracle/_ref/libptmi_ref.so (oracle/Makefile).  Bit-exact:
BVHBuilder output, Scene::intersect (BVH and linear), Sensor, unit_vector,
constructor normals.  Skipped where oracle/_ref was never built.
"""
import os

import numpy as np
import pytest

from oracle_binding import (OracleScene, RefScene, SCENES, Camera, camera_frame, camera_ray, default_camera,
                            ref_available, ref_camera, ref_lib, ref_obj_available, ref_obj_load)

pytestmark = pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (needs /root/reference)")
F = np.float32


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def random_soup(rng, n, quad_frac=0.3, scale=4.0):
    types = (rng.random(n) < quad_frac).astype(np.int32)
    centers = rng.uniform(-scale, scale, (n, 1, 3))
    verts = (centers + rng.normal(0, 0.35, (n, 4, 3))).astype(F)
    # make quads planar-ish parallelograms: v01 = v00 + (v11 - v10)
    q = types == 1
    verts[q, 2] = verts[q, 1] + (verts[q, 3] - verts[q, 0])
    normal = rng.normal(0, 1, (n, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    bsdf = rng.uniform(0, 1, (n, 3)); Le = rng.uniform(0, 2, (n, 3)) * (rng.random((n, 1)) < 0.1)
    return types, verts, normal.astype(F), bsdf.astype(F), Le.astype(F)


def scene_pairs():
    out = []
    for name, kw in [("cbox.obj", {}), ("cbox_quads.obj", {}), ("cbox_quads.obj", dict(convert_quads=True)),
                     ("cbox.obj", dict(subdivision=2)), ("cbox_quads.obj", dict(subdivision=1)),
                     ("cbox_quads.obj", dict(subdivision=2, convert_quads=True))]:
        o = OracleScene.load(SCENES + "/" + name, **kw)
        p = o.prims()
        out.append((f"{name}{kw}", o, RefScene(p["type"], p["verts"], p["normal"], p["bsdf"], p["Le"])))
    rng = np.random.default_rng(7)
    for n in (1, 3, 5, 64, 1500):
        arrs = random_soup(rng, n)
        out.append((f"soup{n}", OracleScene.from_arrays(*arrs), RefScene(*arrs)))
    # degenerate: many coincident centroids -> "extent < 1e-6" leaf with count > 4, and the
    # empty-partition fallback (bvh.h:185-189, 203-206)
    t = np.zeros(9, np.int32); v = np.tile(rng.normal(0, 1, (1, 4, 3)).astype(F), (9, 1, 1))
    nr = np.tile(np.array([[0, 0, 1]], F), (9, 1)); b = np.full((9, 3), 0.5, F); le = np.zeros((9, 3), F)
    out.append(("coincident9", OracleScene.from_arrays(t, v, nr, b, le), RefScene(t, v, nr, b, le)))
    return out


@pytest.fixture(scope="module")
def pairs():
    return scene_pairs()


def test_bvh_builder_bit_exact(pairs):
    for name, o, r in pairs:
        bo, br = o.bvh(), r.bvh()
        assert len(bo["left"]) == len(br["left"]), name
        for k in ("left", "count", "indices"):
            assert (bo[k] == br[k]).all(), (name, k)
        inner = bo["count"] == 0
        assert (bo["right"][inner] == br["right"][inner]).all(), name
        assert (bits(bo["bmin"]) == bits(br["bmin"])).all() and (bits(bo["bmax"]) == bits(br["bmax"])).all(), name


def _rays_for(o, rng, n):
    """Mix of camera rays, random rays, axis-parallel rays (inv_dir = +-inf), rays starting on
    surfaces (t_min edge) and rays aimed at shared edges/vertices (tie cases)."""
    p = o.prims()
    cf = camera_frame(default_camera(), 64, 64)
    os_, ds = [], []
    for _ in range(n // 4):
        a, b = camera_ray(cf, F(rng.random()), F(rng.random())); os_.append(a); ds.append(b)
    lo = p["verts"].reshape(-1, 3).min(0) - 1; hi = p["verts"].reshape(-1, 3).max(0) + 1
    for _ in range(n // 4):
        os_.append(rng.uniform(lo, hi).astype(F)); d = rng.normal(0, 1, 3); ds.append((d / np.linalg.norm(d)).astype(F))
    for _ in range(n // 8):
        os_.append(rng.uniform(lo, hi).astype(F)); d = np.zeros(3, F); d[rng.integers(3)] = rng.choice([-1.0, 1.0]); ds.append(d)
    for _ in range(n // 8):  # -0.0 components
        os_.append(rng.uniform(lo, hi).astype(F)); d = np.array([-0.0, -0.0, -0.0], F); d[rng.integers(3)] = rng.choice([-1.0, 1.0]); ds.append(d)
    nprim = len(p["type"])
    for _ in range(n // 4):  # from a random point toward a vertex / edge midpoint of a random primitive
        i = rng.integers(nprim); nv = 3 if p["type"][i] == 0 else 4
        a = p["verts"][i, rng.integers(nv)]; b = p["verts"][i, rng.integers(nv)]
        target = (a + b) * F(0.5) if rng.random() < 0.5 else a
        org = rng.uniform(lo, hi).astype(F); d = (target - org).astype(np.float64); d /= max(np.linalg.norm(d), 1e-12)
        os_.append(org); ds.append(d.astype(F))
    return np.array(os_, F), np.array(ds, F)


def test_scene_intersect_bit_exact(pairs):
    rng = np.random.default_rng(11)
    total = hits = 0
    for name, o, r in pairs:
        O, D = _rays_for(o, rng, 400)
        for use_bvh in (True, False):
            for (t_min, t_max) in ((1e-4, np.finfo(F).max), (0.5, 6.0)):
                rh = r.intersect(O, D, t_min, t_max, use_bvh)
                for i in range(len(O)):
                    oh = o.intersect(O[i], D[i], t_min, t_max, use_bvh)
                    assert oh.hit == rh[i].hit and oh.prim == rh[i].prim, (name, i, use_bvh)
                    total += 1
                    if oh.hit:
                        hits += 1
                        assert F(oh.t).view(np.uint32) == F(rh[i].t).view(np.uint32), (name, i)
                        for k in ("p", "n", "bsdf", "Le"):
                            assert (bits(list(getattr(oh, k))) == bits(list(getattr(rh[i], k)))).all(), (name, i, k)
    assert hits > total // 5


def test_sensor_bit_exact():
    cams = [default_camera(),
            Camera((0.5, 3.0, 8.5), (0, 2.5, 0), (0, 1, 0), 40.0, 37.5, -12.25, 1),
            Camera((0.5, 3.0, 8.5), (0, 2.5, 0), (0, 1, 0), 55.0, 123.0, 20.0, 1),
            Camera((1.0, 2.0, 7.0), (0.2, 2.5, -1), (0, 1, 0), 30.0, 90.0, 0.0, 0),
            Camera((-2.0, 4.0, 3.0), (0, 2.5, -3), (0.1, 1, 0), 70.0, 200.0, -45.0, 1)]
    for cam in cams:
        for (w, h) in ((1024, 1024), (1920, 1080), (200, 333)):
            mine = camera_frame(cam, w, h).as_array(); ref = ref_camera(cam, w, h)
            assert (bits(mine) == bits(ref)).all(), (list(cam.origin), w, h, mine, ref)
            cf = camera_frame(cam, w, h)
            rng = np.random.default_rng(3)
            for _ in range(50):
                u, v = F(rng.random()), F(rng.random())
                o, d = camera_ray(cf, u, v)
                ro = np.zeros(3, F); rd = np.zeros(3, F)
                ref_lib().ref_camera_ray(ref.ctypes.data, u, v, ro.ctypes.data, rd.ctypes.data)
                assert (bits(o) == bits(ro)).all() and (bits(d) == bits(rd)).all()


def test_constructor_normals_bit_exact():
    """Triangle 4-arg / Quad constructors (triangle.h:23-31, quad.h:23-31) drive convert_quads and
    subdivision normals; compare through a converted + subdivided load."""
    L = ref_lib()
    o = OracleScene.load(SCENES + "/cbox_quads.obj", subdivision=1, convert_quads=True)
    p = o.prims()
    assert len(p["type"]) == 16 * 2 * 4 and (p["type"] == 0).all()
    for i in range(len(p["type"])):
        out = np.zeros(3, F)
        v = p["verts"][i]
        L.ref_tri_geometric_normal(v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data, out.ctypes.data)
        assert (bits(out) == bits(p["normal"][i])).all(), i
    o = OracleScene.load(SCENES + "/cbox_quads.obj", subdivision=2)
    p = o.prims()
    assert len(p["type"]) == 16 * 16 and (p["type"] == 1).all()
    for i in range(len(p["type"])):
        out = np.zeros(3, F)
        v = np.ascontiguousarray(p["verts"][i])
        L.ref_quad_geometric_normal(v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data, v[3].ctypes.data, out.ctypes.data)
        assert (bits(out) == bits(p["normal"][i])).all(), i


def test_large_tessellated_scene_bvh_and_intersect():
    """65,536 displaced triangles (the 1 M-triangle stress scene at 1/16 density): the reference's BVHBuilder and
    Scene::intersect on its 4364-byte Primitive records vs the oracle, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-pathtracer_amd", "python"))
    import ptmi_scenes
    base = OracleScene.load(SCENES + "/cbox_quads.obj").prims()
    sc = ptmi_scenes.tessellated_cornell(base, 64, 32, seed=1)
    assert len(sc["type"]) == 65536
    args = (sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
    o = OracleScene.from_arrays(*args); r = RefScene(*args)
    bo, br = o.bvh(), r.bvh()
    assert len(bo["left"]) == len(br["left"])
    for k in ("left", "count", "indices"):
        assert (bo[k] == br[k]).all(), k
    assert (bits(bo["bmin"]) == bits(br["bmin"])).all() and (bits(bo["bmax"]) == bits(br["bmax"])).all()
    O, D = _rays_for(o, np.random.default_rng(21), 1200)
    rh = r.intersect(O, D)
    hits = 0
    for i in range(len(O)):
        oh = o.intersect(O[i], D[i])
        assert oh.hit == rh[i].hit and oh.prim == rh[i].prim, i
        if oh.hit:
            hits += 1
            assert F(oh.t).view(np.uint32) == F(rh[i].t).view(np.uint32)
    assert hits > 300


def test_radiosity_prepass_primitive_helpers_bit_exact(pairs):
    """The pieces of the radiosity pre-pass that live in primitive.h - getArea (as the constructors
    computed it), centroid, sampleUniform (triangle barycentric / quad split by area ratio) - against the compiled
    reference.  form_factors.h itself needs <cuda_runtime.h> and is restated only."""
    L = ref_lib()
    rng = np.random.default_rng(21)
    for name, o, r in pairs:
        n = o.n_prims
        for i in (range(n) if n <= 64 else rng.integers(0, n, 64)):
            i = int(i)
            area, cen = o.prim_geometry(i)
            assert bits(area) == bits(F(L.ref_area(r.h, i))), (name, i)
            ref_c = np.zeros(3, F); L.ref_centroid(r.h, i, ref_c.ctypes.data)
            assert (bits(cen) == bits(ref_c)).all(), (name, i)
            us = rng.random((12, 2)).astype(F)
            us[0] = [1.0, 1.0]; us[1] = [2.3283064e-10 / 2, 0.5]; us[2] = [0.5, 1.0]      # curand_uniform's range is (0, 1]
            for r1, r2 in us:
                ref_p = np.zeros(3, F); L.ref_sample_uniform(r.h, i, float(r1), float(r2), ref_p.ctypes.data)
                assert (bits(o.sample_uniform(i, r1, r2)) == bits(ref_p)).all(), (name, i, r1, r2)


def test_render_config_layout_and_constants():
    """rendering/render_config.h as the reference's own compiler pass sees it (it reaches ref_harness.cpp through
    triangle.h:7): sizeof / field offsets of PrecomputedCDF (:24-31), GRID_* (:7-9), the SamplingMode values (:38-44), the
    launch block (:51-52) and the grid-step macros (:14-17, doubles because M_PI is the double one) - against the oracle's
    restatement AND the product's record layout (ptmi_host_cdf_record_layout = csrc/device_scene.h kCdf*)."""
    import ctypes as C
    from oracle_binding import oracle_lib
    import ptmi
    ref = np.zeros(22, np.int32); ref_lib().ref_layout(ref.ctypes.data)
    assert ref[:7].tolist() == [2120, 0, 1024, 1056, 1088, 2112, 2116]          # "PrecomputedCDF 2120 B"
    assert ref[7:10].tolist() == [16, 256, 8]
    assert ref[10:15].tolist() == [0, 1, 2, 3, 4]                                # ptmi_config.sampling_mode values
    assert ref[15:17].tolist() == [16, 16]
    assert ref[17:20].tolist() == [4364, 36, 96]                                 # Primitive, BVHNode, SurfaceInteractionRecord
    assert ref[21] == 8                                                          # GRID_D_THETA is a double expression
    orc = np.zeros(10, np.int32); oracle_lib().po_cdf_layout(orc.ctypes.data)
    assert orc.tolist() == ref[:10].tolist()
    assert ptmi.host_cdf_record_layout().tolist() == ref[:10].tolist()
    rc = np.zeros(6, np.float64); ref_lib().ref_grid_constants(rc.ctypes.data)
    oc = np.zeros(6, np.float64); oracle_lib().po_grid_constants(oc.ctypes.data)
    assert (rc.view(np.uint64) == oc.view(np.uint64)).all(), (rc, oc)
    assert rc[4] == np.pi                                                        # M_PI: the double constant, not math_utils.h's float


@pytest.mark.skipif(not ref_obj_available(), reason="oracle/_ref/libptmi_ref_obj.so not built (needs /root/reference and NVIDIA's cuda_runtime.h)")
@pytest.mark.parametrize("name,n_tris,n_quads,warnings", [("cbox.obj", 32, 0, 0), ("cbox_quads.obj", 0, 16, 20)])
def test_obj_loader_against_the_compiled_reference_loader(name, n_tris, n_quads, warnings):
    """loadOBJ + loadMTL of the reference itself (utils/file_manager.h:39-79, 93-273, compiled unmodified against
    NVIDIA's own cuda_runtime.h) on the two Cornell files: the oracle's C loader and the product's C++ loader return its
    primitives bit for bit - type, vertices, stored normal (first vn or geometric), Kd, Ke - incl. the 20 "malformed face
    vertex token" warnings of cbox_quads.obj:100-112 (trailing '# Top' comments), which must not change the result."""
    import ptmi
    path = os.path.join(SCENES, name)
    want = ref_obj_load(path)
    assert want is not None and ((want["type"] == 0).sum(), (want["type"] == 1).sum(), want["warnings"]) == (n_tris, n_quads, warnings)
    for got in (OracleScene.load(path).prims(), ptmi.HostScene.load(path).prims()):
        assert (got["type"] == want["type"]).all()
        tri = got["type"] == 0
        assert (bits(got["verts"][tri][:, :3]) == bits(want["verts"][tri][:, :3])).all() and (bits(got["verts"][~tri]) == bits(want["verts"][~tri])).all()
        for k in ("normal", "bsdf", "Le"):
            assert (bits(got[k]) == bits(want[k])).all(), k
