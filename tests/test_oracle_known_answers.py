"""Pins the oracle against the known answers SURVEY.md §8c recorded from the
reference source (measured by the survey stage with the reference's own
loadOBJ -> BVHBuilder -> Scene::intersect): primitive and node counts, the full
BVH dump of cbox.obj, the default camera vectors and six camera-ray hits.
These exercise the RESTATED OBJ/MTL loader (file_manager.h:39-273) end to end,
because every number below depends on what the loader produced.
"""
import numpy as np

from oracle_binding import (OracleScene, SCENES, camera_frame, camera_ray, default_camera)

F = np.float32


def test_cbox_counts_and_bvh_dump():
    s = OracleScene.load(SCENES + "/cbox.obj")
    assert s.n_prims == 32 and s.n_nodes == 21
    pr = s.prims()
    assert (pr["type"] == 0).all()
    b = s.bvh()
    lrc = list(zip(b["left"].tolist(), b["right"].tolist(), b["count"].tolist()))
    # pre-order (L,R,count); leaves print (start,-,count)
    expect = [(1, 10, 0), (2, 7, 0), (3, 4, 0), (0, -1, 4), (5, 6, 0), (4, -1, 3), (7, -1, 3), (8, 9, 0),
              (10, -1, 2), (12, -1, 4), (11, 20, 0), (12, 13, 0), (16, -1, 3), (14, 17, 0), (15, 16, 0),
              (19, -1, 2), (21, -1, 3), (18, 19, 0), (24, -1, 3), (27, -1, 2), (29, -1, 3)]
    assert lrc == expect
    assert b["indices"].tolist() == [9, 23, 24, 28, 25, 29, 30, 26, 6, 31, 2, 27, 22, 4, 0, 8, 3, 7, 10, 16, 18,
                                     19, 12, 13, 21, 17, 15, 14, 20, 1, 11, 5]
    leaves = [c for (_, _, c) in lrc if c > 0]
    assert len(leaves) == 11 and max(leaves) == 4
    np.testing.assert_allclose(b["bmin"][0], [-3.01401, -0.162687, -5.83997], rtol=0, atol=5e-6)
    np.testing.assert_allclose(b["bmax"][0], [2.54599, 5.32977, -0.243598], rtol=0, atol=5e-6)


def test_cbox_quads_counts():
    s = OracleScene.load(SCENES + "/cbox_quads.obj")   # trailing "# Top" comments are skipped token by token
    assert s.n_prims == 16 and s.n_nodes == 13
    assert (s.prims()["type"] == 1).all()
    assert int((s.bvh()["count"] > 0).sum()) == 7
    # materials come from cbox.mtl (cbox_quads.obj:2), not cbox_quads.mtl
    pr = s.prims()
    assert np.allclose(pr["Le"][0], [25, 25, 25]) and np.allclose(pr["bsdf"][0], [0, 0, 0])
    assert np.allclose(pr["bsdf"][5], [0.0, 0.32, 0.0])


def test_default_camera_vectors():
    cf = camera_frame(default_camera(), 1024, 1024).as_array()
    # SURVEY §8 a3 (values printed with %.9g from the reference's Sensor)
    exp = np.array([-3.72830215e-07, 2.5, 8.52936077, -0.363970578, 2.13602972, 7.52936077,
                    0.72794044, -0.0, 3.18192903e-08, 0.0, 0.72794044, 0.0], F)
    assert (cf == exp).all(), (cf, exp)


def test_camera_ray_known_hits():
    s = OracleScene.load(SCENES + "/cbox.obj")
    cf = camera_frame(default_camera(), 1024, 1024)
    kat = [((0.50, 0.50), 3, 14.3670778), ((0.25, 0.75), 2, 14.8374147), ((0.10, 0.10), 6, 9.87406731),
           ((0.50, 0.93), 4, 9.47220039), ((0.70, 0.30), 20, 10.4547949)]
    for (u, v), prim, t in kat:
        o, d = camera_ray(cf, u, v)
        for use_bvh in (True, False):
            h = s.intersect(o, d, use_bvh=use_bvh)
            assert h.hit == 1 and h.prim == prim and F(h.t) == F(t), ((u, v), h.prim, h.t)
    o, d = camera_ray(cf, 0.70, 0.30)
    h = s.intersect(o, d)
    np.testing.assert_allclose(list(h.n), [0.9578, 0.0002, 0.2873], atol=5e-5)
    o, d = camera_ray(cf, 0.02, 0.50)
    assert s.intersect(o, d).hit == 0


def test_workload_counters_match_survey():
    """rays/sample 2.64 (max_depth 4), nodes/ray 14.0, prim tests/ray 15.5 on cbox.obj.
    (RNG differs from the survey's placeholder, so only to Monte-Carlo accuracy.)"""
    s = OracleScene.load(SCENES + "/cbox.obj")
    _, rad, st = s.render(default_camera(), 128, 128, 16, max_depth=4)
    assert abs(st.rays / st.samples - 2.64) < 0.03
    assert abs(st.node_visits / st.rays - 14.0) < 0.3
    assert abs(st.prim_tests / st.rays - 15.5) < 0.3
    assert abs(float(rad.mean()) - 0.1863) < 0.01


def test_oracle_reproduces_committed_golden_frames():
    """tests/golden/frame_*.npz were written by tests/golden/make_golden.py; the oracle must not drift from them."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_*.npz")))
    assert len(files) >= 5
    c1 = [f for f in files if "256x256_16spp_d4" in f]
    assert c1, "the C1 fixture (BASELINE.json configs[0]) is missing"
    for f in files[:3] + c1:
        g = np.load(f)
        o = OracleScene.load(os.path.join(SCENES, str(g["scene"])), int(g["subdivision"]), bool(g["convert_quads"]))
        rgb, rad, st = o.render(default_camera(), int(g["width"]), int(g["height"]), int(g["spp"]), max_depth=int(g["max_depth"]))
        assert (rad.view(np.uint32) == g["radiance"].view(np.uint32)).all() and (rgb == g["rgb8"]).all(), f
        assert [st.samples, st.rays, st.node_visits, st.prim_tests, st.hits] == g["counters"].tolist()


def test_config1_cbox_256_16spp_depth4_mean_radiance():
    """BASELINE.json configs[0] at its stated size on the CPU build (the oracle): cbox.obj 256x256, 16 spp, max 4
    bounces.  SURVEY 4 item 3 measured mean per-channel L = 0.186291 on the compiled reference with a placeholder RNG, so
    the comparison is statistical: the spread of this mean over RNG seeds is 1.0e-3 (seeds 1, 7, 99, 12345 give
    0.18755, 0.18595, 0.18576, 0.18756), tolerance 3 sigma.  The committed fixture pins the exact frame."""
    s = OracleScene.load(SCENES + "/cbox.obj")
    _, rad, st = s.render(default_camera(), 256, 256, 16, max_depth=4)
    assert st.samples == 256 * 256 * 16
    assert abs(float(rad.mean(dtype=np.float64)) - 0.186291) < 3.0e-3
    assert abs(st.rays / st.samples - 2.64) < 0.01
