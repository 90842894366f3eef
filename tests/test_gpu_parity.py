"""GPU parity: the HIP path (through the C ABI of libptmi.so) against the CPU oracle, bit for bit.

Tolerance: NONE for linear float radiance and for the 8-bit image - the path is IEEE binary32 with a
shared numerics contract (include/ptmi_math.h), so the expected number of differing pixels is 0 and
the tests assert exactly that (BASELINE.json's bar is RMSE < 1e-4; RMSE here must be exactly 0).
"""
import os

import numpy as np
import pytest

import ptmi
from oracle_binding import (OracleScene, SCENES, Camera as OCamera, camera_frame, camera_ray, default_camera,
                            oracle_lib, rng_stream)

from guided_fixtures import synthetic_radiosity_grids

pytestmark = pytest.mark.gpu
F = np.float32
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


@pytest.fixture(scope="module")
def R():
    r = ptmi.Renderer(0)
    yield r
    r.close()


def ocam_of(cam):
    return OCamera(tuple(cam.origin), tuple(cam.lookat), tuple(cam.vup), cam.vfov_deg, cam.yaw_deg, cam.pitch_deg, cam.orbit)


def full_frame(r):
    rgb, rad = r.read_image()
    rows = r.local_rows()
    return rows, rgb, rad


def assert_same_counters(R, st, ost):
    """The workload counters are the ORACLE's, exactly, whenever the walk visits the reference's tree.  The certified walk (the
    automatic choice of triangle scenes above 64 primitives) visits another tree and proves that it finds the reference's hits:
    rays and hits must still be the oracle's, its own node / test counts are smaller; forced onto the reference's tree the same
    frame then gives the oracle's counts (test_every_traversal_mode_gives_the_same_frame, test_packed_layout_...)."""
    assert (st.rays, st.hits) == (ost.rays, ost.hits)
    if R.traversal() == R.CERTIFIED:
        assert st.node_visits < ost.node_visits and st.cert_fallback <= st.cert_chain + st.hits // 100 + 8
    else:
        assert (st.node_visits, st.prim_tests) == (ost.node_visits, ost.prim_tests)


def assert_same_image(rgb, rad, orgb, orad, what=""):
    nd = int((bits(rad) != bits(orad)).any(axis=-1).sum())
    rmse = float(np.sqrt(np.mean((rad.astype(np.float64) - orad.astype(np.float64)) ** 2)))
    assert nd == 0 and rmse == 0.0, f"{what}: {nd} pixels differ, RMSE {rmse:.3e}"
    assert (rgb == orgb).all(), f"{what}: rgb8 differs in {(rgb != orgb).any(axis=-1).sum()} pixels"


# ------------------------------------------------------------------------------------------------
# single stages
# ------------------------------------------------------------------------------------------------
def test_rng_streams_match_oracle(R):
    pixels = np.array([0, 1, 2, 63, 64, 1023, 1024, 65535, 1024 * 1024 - 1, 4096 * 4096 - 1, 123457], np.int32)
    got = R.debug_rng(2023, pixels, 16)
    for i, p in enumerate(pixels):
        exp, _ = rng_stream(2023 + int(p), int(p), 16)
        assert (bits(got[i]) == bits(exp)).all(), p


def test_cosine_sampling_matches_oracle(R):
    rng = np.random.default_rng(1)
    n = 20000
    nr = rng.normal(0, 1, (n, 3)); nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    nr = nr.astype(F)
    nr[:10] = [0, 0, -1]                      # Frisvad special case (integrator.h:74-76)
    nr[10:20] = [0, 1, 0]
    u = rng.random(n).astype(F); v = rng.random(n).astype(F)
    u[20:24] = [1.0, 1.0, 2.3283064e-10 / 2, 0.5]; v[20:24] = [1.0, 0.25, 0.75, 1.0]
    got = R.debug_cosine_sample(nr, u, v)
    L = oracle_lib()
    exp = np.zeros((n, 3), F)
    for i in range(n):
        L.po_sample_cosine_hemisphere(nr[i].ctypes.data, float(u[i]), float(v[i]), exp[i].ctypes.data)
    assert (bits(got) == bits(exp)).all(), int((bits(got) != bits(exp)).any(axis=1).sum())


def test_fast_reciprocal_is_exact(R):
    """The Moller-Trumbore determinant is inverted with 1 v_rcp_f32 + 4 fma instead of the 10-instruction IEEE
    division.  Exhaustive: every normal float whose reciprocal is normal too (both signs) must give the IEEE bits."""
    for sign in (0, 0x80000000):
        bad, first = R.debug_rcp_check(sign + (1 << 23), (253 - 1) << 23)      # exponents 1..252: 2^-126 <= |a| < 2^126
        assert bad == 0, (bad, hex(first))
    bad, _ = R.debug_rcp_check(0, 1 << 23)                                     # denormal inputs are NOT covered ...
    assert bad > 0                                                             # ... and the test can tell


def _test_rays(o, rng, n):
    p = o.prims()
    cf = camera_frame(default_camera(), 64, 64)
    os_, ds = [], []
    for _ in range(n // 2):
        a, b = camera_ray(cf, F(rng.random()), F(rng.random())); os_.append(a); ds.append(b)
    lo = p["verts"].reshape(-1, 3).min(0) - 1; hi = p["verts"].reshape(-1, 3).max(0) + 1
    for _ in range(n // 4):
        os_.append(rng.uniform(lo, hi).astype(F)); d = rng.normal(0, 1, 3); ds.append((d / np.linalg.norm(d)).astype(F))
    for _ in range(n // 8):
        os_.append(rng.uniform(lo, hi).astype(F)); d = np.array([-0.0, 0.0, -0.0], F); d[rng.integers(3)] = rng.choice([-1.0, 1.0]); ds.append(d)
    nprim = len(p["type"])
    for _ in range(n // 8):
        i = rng.integers(nprim); nv = 3 if p["type"][i] == 0 else 4
        a = p["verts"][i, rng.integers(nv)]; b = p["verts"][i, rng.integers(nv)]
        org = rng.uniform(lo, hi).astype(F); d = ((a + b) * F(0.5) - org).astype(np.float64); d /= max(np.linalg.norm(d), 1e-12)
        os_.append(org); ds.append(d.astype(F))
    return np.array(os_, F), np.array(ds, F)


@pytest.mark.parametrize("name,sub,conv", [("cbox.obj", 0, False), ("cbox_quads.obj", 0, False), ("cbox_quads.obj", 2, True),
                                           ("cbox_quads.obj", 3, False), ("cbox.obj", 3, False)])
def test_scene_intersect_matches_oracle(R, name, sub, conv):
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    o = OracleScene.load(path, sub, conv)
    O, D = _test_rays(o, np.random.default_rng(5), 2000)
    for (t_min, t_max) in ((1e-4, float(np.finfo(F).max)), (0.5, 6.0)):
        g = R.debug_intersect(O, D, t_min, t_max)
        nh = 0
        for i in range(len(O)):
            h = o.intersect(O[i], D[i], t_min, t_max)
            assert g["hit"][i] == h.hit and g["prim"][i] == h.prim, (i, g["prim"][i], h.prim)
            if h.hit:
                nh += 1
                assert bits(g["t"][i]) == bits(F(h.t))
                assert (bits(g["p"][i]) == bits(list(h.p))).all() and (bits(g["n"][i]) == bits(list(h.n))).all()
        assert nh > 300


# ------------------------------------------------------------------------------------------------
# whole frames
# ------------------------------------------------------------------------------------------------
FRAMES = [  # scene, sub, conv, W, H, spp, max_depth
    ("cbox.obj", 0, False, 64, 64, 4, 5),
    ("cbox.obj", 0, False, 128, 128, 16, 4),
    ("cbox.obj", 0, False, 96, 64, 8, 8),
    ("cbox_quads.obj", 0, False, 128, 72, 16, 5),
    ("cbox_quads.obj", 0, True, 64, 64, 8, 8),
    ("cbox.obj", 1, False, 64, 64, 8, 5),         # 128 triangles: PHASED walk, scene LDS-resident
    ("cbox.obj", 2, False, 64, 64, 8, 5),         # 512 triangles: PHASED walk from L2
    ("cbox_quads.obj", 3, False, 64, 64, 4, 5),   # 1024 quads: PHASED walk from L2
    ("cbox.obj", 0, False, 33, 17, 3, 1),         # ragged size, depth 1
    ("cbox.obj", 0, False, 1, 1, 5, 5),           # single pixel
]


@pytest.mark.parametrize("name,sub,conv,W,H,spp,depth", FRAMES)
def test_frame_matches_oracle(R, name, sub, conv, W, H, spp, depth):
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    R.update_resolution(W, H)
    R.set_config(spp=spp, max_depth=depth, segments_per_launch=0, collect_stats=True)
    st = R.render_frame()
    rgb, rad = R.read_image()
    o = OracleScene.load(path, sub, conv)
    orgb, orad, ost = o.render(default_camera(), W, H, spp, max_depth=depth)
    assert_same_image(rgb, rad, orgb, orad, f"{name} {W}x{H}")
    # the workload counters feeding the roofline model are the oracle's, exactly
    assert st.samples == ost.samples
    assert_same_counters(R, st, ost)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("name,sub,conv", [("cbox.obj", 0, False), ("cbox_quads.obj", 0, False), ("cbox.obj", 1, False)])
def test_every_traversal_mode_gives_the_same_frame(R, mode, name, sub, conv):
    """SWEEP (wave-uniform), LANE (stackless per lane), STACK (explicit stack) and PHASED (wave-scheduled phases)
    must agree bit for bit."""
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    assert R.set_traversal(mode) == mode
    try:
        W, H, spp = 72, 40, 6
        R.update_resolution(W, H)
        R.set_config(spp=spp, max_depth=5, collect_stats=True)
        st = R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad, ost = OracleScene.load(path, sub, conv).render(default_camera(), W, H, spp, max_depth=5)
        assert_same_image(rgb, rad, orgb, orad, f"{name} mode {mode}")
        assert_same_counters(R, st, ost)
        O, D = _test_rays(OracleScene.load(path, sub, conv), np.random.default_rng(9), 600)
        g = R.debug_intersect(O, D)
        o = OracleScene.load(path, sub, conv)
        for i in range(len(O)):
            h = o.intersect(O[i], D[i])
            assert g["hit"][i] == h.hit and g["prim"][i] == h.prim and (not h.hit or bits(g["t"][i]) == bits(F(h.t)))
    finally:
        R.set_traversal(-1)


@pytest.mark.parametrize("name,sub,conv,sampling", [("cbox.obj", 3, False, 0), ("cbox_quads.obj", 4, False, 0), ("cbox_quads.obj", 3, True, 3),
                                                    ("cbox.obj", 5, False, 0), ("cbox.obj", 4, False, 1)])
def test_packed_layout_gives_the_same_frame(R, name, sub, conv, sampling):
    """PACKED: the phased walk over sibling-pair node records with explicit links, 36-byte triangle records and the material
    table.  Frame and workload counters must equal the oracle's and the pre-order PHASED walk's - triangles and native quads,
    guided sampling (records reached through load_index), counters on/off, several segments-per-launch."""
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    o = OracleScene.load(path, sub, conv)
    try:
        assert R.set_packed_min_nodes(16) > R.scene_info()["n_bvh_nodes"]          # pairs: root + padding + 2 per inner node
        # the automatic choice for a triangle scene with the fast tree built is the certified walk; this test is about the packed one
        assert R.set_traversal(-1) in (R.PACKED, R.CERTIFIED) and R.set_traversal(R.PACKED) == R.PACKED
        if sampling:
            grids = synthetic_radiosity_grids(o.n_prims, seed=sampling)
            R.set_radiosity_grids(grids); o.set_radiosity_grids(grids); o.set_mis_fraction(0.5)
        W, H, spp = 96, 50, 5
        orgb, orad, ost = o.render(default_camera(), W, H, spp, max_depth=6, sampling_mode=sampling)
        for stats in (True, False):
            for seg in (0, 1, 5):
                R.update_resolution(W, H)
                R.set_config(spp=spp, max_depth=6, collect_stats=stats, segments_per_launch=seg, sampling_mode=sampling, mis_bsdf_fraction=0.5)
                st = R.render_frame()
                rgb, rad = R.read_image()
                assert_same_image(rgb, rad, orgb, orad, f"{name} packed stats {stats} seg {seg}")
                if stats:
                    assert (st.rays, st.node_visits, st.prim_tests, st.hits) == (ost.rays, ost.node_visits, ost.prim_tests, ost.hits)
        for top in (0, 6, 64, 2048):                                    # how much of the tree is walked from LDS must not matter
            n_top, d_top = R.set_packed_top(top)
            assert n_top <= max(top, 0) and (n_top == 0) == (top < 4)
            assert R.set_traversal(R.PACKED) == R.PACKED
            R.update_resolution(W, H); R.set_config(collect_stats=True); st = R.render_frame()
            assert_same_image(*R.read_image(), orgb, orad, f"{name} packed, LDS top {top}")
            assert (st.rays, st.node_visits, st.prim_tests, st.hits) == (ost.rays, ost.node_visits, ost.prim_tests, ost.hits)
            # every ray visits the root (position 0 < n_top); nothing is served from LDS without a top
            assert (st.top_node_visits == 0) if n_top == 0 else (st.rays <= st.top_node_visits <= st.node_visits)
            if n_top >= R.scene_info()["n_bvh_nodes"] + 1: assert st.top_node_visits == st.node_visits
        R.set_config(collect_stats=False)
        assert R.set_traversal(R.PHASED) == R.PHASED
        R.update_resolution(W, H); R.render_frame()
        assert_same_image(*R.read_image(), orgb, orad, "phased")
        assert R.set_packed_min_nodes(1 << 30) == 0 and R.set_traversal(-1) == R.CERTIFIED   # no packed layout: the certified walk stays the automatic choice
        assert R.set_traversal(R.PACKED) == R.PHASED
    finally:
        R.set_traversal(-1); R.set_packed_min_nodes(8192); R.set_packed_top(512)
        R.set_config(sampling_mode=0, segments_per_launch=0, collect_stats=False)


def test_deep_tree_uses_the_stack_walk_and_the_references_drop_rule(R):
    """Centroids at 2^-i make the midpoint split peel one primitive per level: depth > 62, where the reference
    silently drops children once its 64-entry stack holds 62 (scene.h:101-105).  Oracle and GPU must drop the same."""
    n = 100
    x = (2.0 ** 90 * 2.2 ** (-np.arange(n, dtype=np.float64))).astype(F)
    verts = np.zeros((n, 4, 3), F)
    for i in range(n):
        z = F(i) * F(0.01)
        verts[i, 0] = [x[i], -0.004, z]; verts[i, 1] = [x[i], 0.004, z]; verts[i, 2] = [x[i], 0.0, z + F(0.008)]
    types = np.zeros(n, np.int32)
    nr = np.tile(np.array([[1, 0, 0]], F), (n, 1)); b = np.full((n, 3), 0.5, F); le = np.ones((n, 3), F)
    R.load_scene_arrays(types, verts, nr, b, le)
    info = R.scene_info()
    assert info["bvh_depth"] > 62
    assert R.set_traversal(-1) == R.STACK and R.set_traversal(R.LANE) == R.STACK     # cannot be forced away
    o = OracleScene.from_arrays(types, verts, nr, b, le)
    rng = np.random.default_rng(2)
    N = 800
    O = np.stack([np.full(N, 2.0 ** 91), rng.uniform(-0.001, 0.001, N), rng.uniform(0, n * 0.01, N)], 1).astype(F)
    D = np.tile(np.array([[-1, 0, 0]], F), (N, 1))
    g = R.debug_intersect(O, D)
    dropped = 0
    for i in range(N):
        h = o.intersect(O[i], D[i]); hl = o.intersect(O[i], D[i], use_bvh=False)
        assert g["hit"][i] == h.hit and g["prim"][i] == h.prim
        dropped += int(h.prim != hl.prim)
    assert dropped > 0          # the drop rule really changes answers here, and both sides reproduce it
    R.set_traversal(-1)


def test_golden_fixtures(R):
    """Committed fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle)."""
    import glob
    files = sorted(glob.glob(os.path.join(GOLDEN, "frame_*.npz")))
    assert files, "no golden fixtures"
    for f in files:
        g = np.load(f)
        name = str(g["scene"]); W, H, spp, depth = (int(g[k]) for k in ("width", "height", "spp", "max_depth"))
        R.load_scene(os.path.join(SCENES, name), int(g["subdivision"]), bool(g["convert_quads"]))
        R.update_resolution(W, H); R.set_config(spp=spp, max_depth=depth, collect_stats=False)
        R.render_frame()
        rgb, rad = R.read_image()
        assert_same_image(rgb, rad, g["rgb8"], g["radiance"], os.path.basename(f))


def test_result_independent_of_segments_per_launch(R):
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    R.update_resolution(96, 96)
    ref = None
    for seg in (1, 2, 3, 8, 64, 100000):
        R.update_resolution(96, 96)               # re-seeds the streams (allocateBuffers -> render_init)
        R.set_config(spp=8, max_depth=5, segments_per_launch=seg, collect_stats=False)
        st = R.render_frame()
        rgb, rad = R.read_image()
        if ref is None:
            ref = (rgb, rad)
        else:
            assert (bits(rad) == bits(ref[1])).all() and (rgb == ref[0]).all(), seg
        if seg == 100000:
            assert st.bounce_launches <= 2        # megakernel limit: one launch does the whole frame (+ one run-ahead launch that finds the queue empty)


def test_result_independent_of_wave_pixel_layout(R):
    """64x1 strips per wave (default) vs 8x8 pixel tiles: scheduling only."""
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    out = []
    for strips in (0, 1):
        R.set_config(spp=6, max_depth=5, wave_tiles=strips)
        R.update_resolution(128, 64)
        R.render_frame()
        out.append(R.read_image())
    assert (bits(out[0][1]) == bits(out[1][1])).all() and (out[0][0] == out[1][0]).all()
    orgb, orad, _ = OracleScene.load(os.path.join(SCENES, "cbox.obj")).render(default_camera(), 128, 64, 6)
    assert_same_image(out[1][0], out[1][1], orgb, orad, "tile8")
    R.set_config(wave_tiles=0)


def test_result_independent_of_stream_chunks(R):
    """One stream vs two independent pixel chunks on two streams: scheduling only."""
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    out = []
    for streams in (1, 2):
        R.set_config(spp=5, max_depth=5, streams=streams, segments_per_launch=3, collect_stats=True)
        R.update_resolution(200, 120)
        st = R.render_frame()
        out.append(R.read_image() + (st,))
    assert (bits(out[0][1]) == bits(out[1][1])).all() and (out[0][0] == out[1][0]).all()
    assert (out[0][2].rays, out[0][2].node_visits, out[0][2].prim_tests) == (out[1][2].rays, out[1][2].node_visits, out[1][2].prim_tests)
    assert out[0][2].path_visits == out[1][2].path_visits
    orgb, orad, _ = OracleScene.load(os.path.join(SCENES, "cbox.obj")).render(default_camera(), 200, 120, 5)
    assert_same_image(out[1][0], out[1][1], orgb, orad, "2 streams")
    R.set_config(streams=0, segments_per_launch=0, collect_stats=False)


def test_rng_state_persists_across_frames(R):
    """No accumulation across frames, only the RNG carries over (integrator.h:379)."""
    W = H = 48
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    R.update_resolution(W, H); R.set_config(spp=3, max_depth=5)
    o = OracleScene.load(os.path.join(SCENES, "cbox.obj"))
    state = np.zeros((H * W, 6), np.uint32)
    for frame in range(3):
        R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad, _ = o.render(default_camera(), W, H, 3, rng_state=state, reset_rng=(frame == 0))
        assert_same_image(rgb, rad, orgb, orad, f"frame {frame}")
    R.update_resolution(W, H)                     # re-seeded: frame 0 again
    R.render_frame()
    _, rad0 = R.read_image()
    _, orad0, _ = o.render(default_camera(), W, H, 3)
    assert (bits(rad0) == bits(orad0)).all()


@pytest.mark.parametrize("name,sub,W,H,spp,depth,n", [("cbox.obj", 0, 96, 64, 5, 6, 4), ("cbox_quads.obj", 0, 50, 31, 3, 5, 3),
                                                      ("cbox.obj", 2, 64, 48, 4, 5, 3), ("cbox.obj", 3, 64, 40, 3, 6, 5)])
def test_frame_batch_equals_successive_frames(R, name, sub, W, H, spp, depth, n):
    """ptmi_render_frames(n): pixels go from frame k straight on to frame k + 1 (pipelined), every frame of the batch must be
    bit-identical to the k-th of n separate ptmi_render_frame calls and to the oracle's k-th frame; the RNG streams afterwards
    stand where n separate calls leave them (one more frame compared)."""
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub)
    if sub == 3:
        R.set_packed_min_nodes(16)
    o = OracleScene.load(path, sub)
    state = np.zeros((H * W, 6), np.uint32)
    want = []
    for k in range(n + 1):
        orgb, orad, _ = o.render(default_camera(), W, H, spp, max_depth=depth, rng_state=state, reset_rng=(k == 0))
        want.append((orgb.copy(), orad.copy()))
    try:
        for seg in (0, 2):
            R.update_resolution(W, H)
            R.set_config(spp=spp, max_depth=depth, segments_per_launch=seg, collect_stats=False)
            st = R.render_frames(n)
            assert st.samples == W * H * spp * n
            assert_same_image(*R.read_image(), *want[n - 1], "last frame of the batch")
            for k in (0, n - 1, 1, n - 2):
                R.select_frame(k)
                assert_same_image(*R.read_image(), *want[k], f"{name} frame {k} of {n} seg {seg}")
            with pytest.raises(ptmi.PtmiError):
                R.select_frame(n)
            R.render_frame()                                        # streams carried on across the batch
            assert_same_image(*R.read_image(), *want[n], "frame after the batch")
            R.select_frame(0)
            assert_same_image(*R.read_image(), *want[n], "a single frame is a batch of one")
        with pytest.raises(ptmi.PtmiError):
            R.render_frames(0)
        R.set_config(spp=65536)
        with pytest.raises(ptmi.PtmiError):
            R.render_frames(2)
    finally:
        R.set_config(spp=1, segments_per_launch=0); R.set_packed_min_nodes(8192)


def test_select_frame_after_a_radiosity_frame_is_rejected(R):
    """ptmi_select_frame resolves the colour sums of the last path-tracing run.  After a Radiosity-integrator frame (or a run
    that failed) those sums belong to some earlier frame: the call must return PTMI_E_INVALID and leave the image alone."""
    path = os.path.join(SCENES, "cbox.obj")
    R.load_scene(path); R.update_resolution(48, 40)
    try:
        R.set_config(spp=4, max_depth=5, integrator=0)
        R.render_frame()
        R.select_frame(0)                                           # fine after a path-tracing frame
        R.set_config(integrator=1)
        R.render_frame()
        rgb, rad = R.read_image()
        with pytest.raises(ptmi.PtmiError) as e:
            R.select_frame(0)
        assert e.value.code == -1
        rgb2, rad2 = R.read_image()
        assert (rgb == rgb2).all() and (bits(rad) == bits(rad2)).all()
    finally:
        R.set_config(integrator=0, spp=1)


@pytest.mark.parametrize("name,sub", [("05_instances.pbrt", 0), ("03_arealight.pbrt", 1), ("04_materials.pbrt", 0)])
def test_pbrt_scene_renders_like_the_oracle_on_the_same_primitives(R, name, sub):
    """ptmi_load_scene on a .pbrt file (own parser, pinned against the compiled reference importer in
    tests/test_pbrt_loader.py), then the ordinary path: the frame must equal the oracle's on the golden primitive arrays."""
    path = os.path.join(GOLDEN, "pbrt", name)
    g = np.load(os.path.splitext(path)[0] + ".npz")
    R.load_scene(path, sub)
    p = R.scene_prims()
    if sub == 0:
        for k in ("verts", "normal", "bsdf", "Le"):
            assert (bits(p[k]) == bits(g[k])).all(), k
    else:
        assert len(p["type"]) == len(g["type"]) * 4 ** sub             # subdivide_primitives after the import, as for OBJ scenes
    o = OracleScene.from_arrays(p["type"], p["verts"], p["normal"], p["bsdf"], p["Le"])
    cam = ptmi.default_camera(); cam.origin[:] = (3.0, 2.5, 14.0); cam.lookat[:] = (0.5, 0.5, 6.0); cam.orbit = 0
    R.set_camera(cam)
    W, H, spp = 72, 48, 6
    R.update_resolution(W, H); R.set_config(spp=spp, max_depth=5, collect_stats=True)
    try:
        st = R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad, ost = o.render(ocam_of(cam), W, H, spp, max_depth=5)
        assert_same_image(rgb, rad, orgb, orad, name)
        assert (st.rays, st.node_visits, st.prim_tests, st.hits) == (ost.rays, ost.node_visits, ost.prim_tests, ost.hits)
        assert st.hits > 0
    finally:
        R.set_camera(ptmi.default_camera()); R.set_config(collect_stats=False)


def test_camera_and_seed_parameters(R):
    cam = ptmi.Camera((0.5, 3.0, 8.5), (0, 2.5, 0), (0, 1, 0), 55.0, 70.0, 10.0, 1)
    path = os.path.join(SCENES, "cbox_quads.obj")
    R.load_scene(path)
    R.set_camera(cam)
    R.set_config(spp=4, max_depth=5, seed_base=777)
    R.update_resolution(80, 45)
    assert (bits(R.camera_frame()) == bits(camera_frame(ocam_of(cam), 80, 45).as_array())).all()
    R.render_frame()
    rgb, rad = R.read_image()
    orgb, orad, _ = OracleScene.load(path).render(ocam_of(cam), 80, 45, 4, seed_base=777)
    assert_same_image(rgb, rad, orgb, orad, "custom camera")
    R.set_camera(ptmi.default_camera()); R.set_config(seed_base=2023)


def test_tile_union_is_the_single_gpu_frame(R):
    """Multi-GPU sharding contract on one device: for every (n_ranks, row_block) the union of the ranks'
    rows is bit-identical to the unsharded frame."""
    W, H, spp = 80, 61, 6
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    R.set_config(spp=spp, max_depth=5)
    R.update_resolution(W, H)
    R.render_frame()
    rgb1, rad1 = R.read_image()
    for n_ranks, rb in ((2, 8), (3, 4), (8, 8), (4, 1)):
        rgb = np.zeros_like(rgb1); rad = np.full_like(rad1, -1)
        for rank in range(n_ranks):
            R.update_resolution(W, H, n_ranks=n_ranks, rank=rank, row_block=rb)
            R.render_frame()
            rows = R.local_rows()
            a, b = R.read_image()
            if len(rows):
                rgb[rows] = a; rad[rows] = b
        assert (bits(rad) == bits(rad1)).all() and (rgb == rgb1).all(), (n_ranks, rb)


def test_errors_are_reported_not_swallowed(R):
    with pytest.raises(ptmi.PtmiError):
        R.load_scene("/nonexistent/scene.obj")
    with pytest.raises(ptmi.PtmiError):
        R.render_frame()                          # previous scene was dropped by the failed load (cleanup first, as the reference)
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    with pytest.raises(ptmi.PtmiError):
        R.update_resolution(0, 10)
    with pytest.raises(ptmi.PtmiError):
        R.set_config(spp=0)
    R.set_config(spp=1)
    # a rejected config changes nothing: spp 7 must not be applied when `streams` is out of range
    R.update_resolution(32, 32); R.set_config(spp=2, max_depth=5)
    R.render_frame(); want = R.read_image()[1].copy()
    R.config.streams = 99
    with pytest.raises(ptmi.PtmiError):
        R.set_config(spp=7)
    R.config.streams = 0; R.config.spp = 2
    R.update_resolution(32, 32)
    st = R.render_frame()
    assert st.samples == 32 * 32 * 2 and (bits(R.read_image()[1]) == bits(want)).all()
    R.set_config(spp=1)


def test_render_frame_downloads_the_image_like_the_reference(R):
    """config.download_image: ptmi_render_frame ends with the D2H of the 8-bit image into the ctx's pinned host image
    (RenderState::h_image, application.h:211)"""
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    R.update_resolution(80, 60); R.set_config(spp=4, max_depth=5, download_image=True)
    R.render_frame()
    host = R.host_image().copy()
    rgb, _ = R.read_image()
    assert host.shape == rgb.shape and (host == rgb).all() and int(host.max()) > 0
    R.set_config(integrator=1)                     # the Radiosity integrator's frame is downloaded too
    R.render_frame()
    assert (R.host_image() == R.read_image()[0]).all()
    R.set_config(integrator=0, download_image=False)


# ------------------------------------------------------------------------------------------------
# guided sampling modes: grid / MIS over per-primitive PrecomputedCDF records
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name,sub,conv,trav", [("cbox.obj", 0, False, -1), ("cbox_quads.obj", 0, False, -1),
                                                ("cbox.obj", 1, False, -1), ("cbox.obj", 0, False, 1), ("cbox.obj", 2, False, 2)])
@pytest.mark.parametrize("mode,frac", [(1, 0.5), (2, 0.5), (3, 0.5), (3, 0.2), (4, 0.5)])
def test_guided_modes_match_oracle(R, name, sub, conv, trav, mode, frac):
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    R.set_traversal(trav)
    o = OracleScene.load(path, sub, conv)
    grids = synthetic_radiosity_grids(o.n_prims, seed=mode)
    R.set_radiosity_grids(grids); o.set_radiosity_grids(grids); o.set_mis_fraction(frac)
    assert (bits(R.precomputed_cdfs()) == bits(o.cdfs())).all()            # host precomputeCDFs == oracle, bit for bit
    W, H, spp, depth = 64, 40, 6, 5
    R.update_resolution(W, H)
    R.set_config(spp=spp, max_depth=depth, sampling_mode=mode, mis_bsdf_fraction=frac, collect_stats=True)
    try:
        st = R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad, ost = o.render(default_camera(), W, H, spp, max_depth=depth, sampling_mode=mode)
        assert_same_image(rgb, rad, orgb, orad, f"{name} mode {mode}")
        assert_same_counters(R, st, ost)
        # and the guided frame really differs from the BSDF frame
        R.update_resolution(W, H); R.set_config(sampling_mode=0)
        R.render_frame()
        assert (bits(R.read_image()[1]) != bits(rad)).any()
    finally:
        R.set_config(sampling_mode=0, mis_bsdf_fraction=0.5, collect_stats=False)
        R.set_traversal(-1)


def test_guided_modes_without_records_equal_bsdf(R):
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    R.update_resolution(64, 64); R.set_config(spp=4, max_depth=5, sampling_mode=0)
    R.render_frame()
    ref = R.read_image()
    for mode in (1, 2, 3, 4):
        R.update_resolution(64, 64); R.set_config(sampling_mode=mode)
        R.render_frame()
        got = R.read_image()
        assert (bits(got[1]) == bits(ref[1])).all() and (got[0] == ref[0]).all(), mode
    R.set_radiosity_grids(np.zeros((32, 256, 3), F))                       # records present, all invalid
    R.update_resolution(64, 64); R.set_config(sampling_mode=3)
    R.render_frame()
    assert (bits(R.read_image()[1]) == bits(ref[1])).all()
    with pytest.raises(ptmi.PtmiError):
        R.set_radiosity_grids(np.zeros((31, 256, 3), F))                   # wrong primitive count
    R.set_radiosity_grids(None)
    R.set_config(sampling_mode=0)


def test_command_line_caller_writes_the_same_frame(R, tmp_path):
    """tools/ptmi_render.py (load -> radiosity pre-pass -> MIS-guided frame -> Save PNG) as a child process: the PNG
    holds the bytes the library returns for the same calls, top row first."""
    import struct, subprocess, sys, zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "cli.png")
    scene = os.path.join(SCENES, "cbox.obj")
    cmd = [sys.executable, os.path.join(root, "tools", "ptmi_render.py"), "--scene", scene, "--width", "96", "--height", "64",
           "--spp", "3", "--subdivision", "1", "--radiosity", "--mc-samples", "8", "--radiosity-steps", "2", "--sampling-mode", "3", "--out", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "128 primitives" in res.stdout and "Msamples/s" in res.stdout
    data = open(out, "rb").read()
    pos = 8; idat = b""; ihdr = None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        if typ == b"IHDR": ihdr = struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n])
        if typ == b"IDAT": idat += data[pos + 8:pos + 8 + n]
        pos += 12 + n
    assert ihdr == (96, 64, 8, 2, 0, 0, 0)
    png = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(64, 1 + 96 * 3)[:, 1:].reshape(64, 96, 3)
    R.load_scene(scene, 1, False)
    R.run_radiosity_solver(mc_samples=8, num_iterations=2)
    R.update_resolution(96, 64); R.set_config(spp=3, max_depth=5, sampling_mode=3)
    R.render_frame()
    rgb, _ = R.read_image()
    assert (png == rgb[::-1]).all()
    R.set_config(sampling_mode=0)


def _random_soup(rng, n, quad_frac=0.35, extent=3.0, size=0.6):
    """triangles and (generally non-planar) quads floating in front of the default camera, a tenth of them emitters"""
    types = (rng.random(n) < quad_frac).astype(np.int32)
    centers = np.stack([rng.uniform(-extent, extent, n), rng.uniform(0.2, 5.0, n), rng.uniform(-5.5, 0.5, n)], 1)[:, None, :]
    verts = (centers + rng.normal(0, size, (n, 4, 3))).astype(F)
    normal = rng.normal(0, 1, (n, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    bsdf = rng.uniform(0.1, 0.95, (n, 3)).astype(F)
    Le = (rng.uniform(0, 6, (n, 3)) * (rng.random((n, 1)) < 0.1)).astype(F)
    return types, verts, normal.astype(F), bsdf, Le


@pytest.mark.parametrize("n,seed,mode", [(1, 8, 0), (2, 9, 3), (7, 1, 0), (60, 2, 0), (64, 3, 3), (65, 4, 0), (400, 5, 2), (3000, 6, 0), (3000, 7, 3)])
def test_random_soup_frames_match_oracle(R, n, seed, mode):
    """Scenes nothing was tuned on: random BVH shapes on both sides of the 64-primitive sweep limit, leaves of mixed
    triangles and skewed quads, stored normals unrelated to the geometry (the loader keeps the file's normals too),
    emitters anywhere; BSDF, grid and MIS sampling (mode) with synthetic radiosity grids."""
    from guided_fixtures import synthetic_radiosity_grids
    rng = np.random.default_rng(seed)
    arrs = _random_soup(rng, n)
    R.load_scene_arrays(*arrs)
    o = OracleScene.from_arrays(*arrs)
    if mode:
        grids = synthetic_radiosity_grids(n, seed=seed)
        R.set_radiosity_grids(grids); o.set_radiosity_grids(grids)
    W, H, spp = 72, 56, 6
    R.update_resolution(W, H)
    R.set_config(spp=spp, max_depth=6, sampling_mode=mode, collect_stats=True)
    st = R.render_frame()
    rgb, rad = R.read_image()
    orgb, orad, ost = o.render(default_camera(), W, H, spp, max_depth=6, sampling_mode=mode)
    assert_same_image(rgb, rad, orgb, orad, f"soup n={n} mode={mode}")
    assert_same_counters(R, st, ost)
    assert n < 7 or (ost.hits > 500 and rad.max() > 0)
    # and the radiosity pre-pass on the same soup (any-hit walk, quad sampling, culling on arbitrary normals)
    if n <= 400:
        R.run_radiosity_solver(mc_samples=6, num_iterations=2)
        got = R.radiosity_solution()
        exp = o.radiosity_solve(mc_samples=6, num_iterations=2)
        for k in ("form_factors", "radiosity", "grid", "radiosity_grid"):
            assert (bits(got[k]) == bits(exp[k])).all(), k
    R.set_config(sampling_mode=0, collect_stats=False)


def _degenerate_scene(rng, quads):
    """the Cornell box plus everything a sloppy exporter produces: zero-area and collinear triangles, duplicated vertices,
    edges of 1e-20 and of 1e6, exactly axis-aligned faces through the camera's view, faces coplanar with box walls,
    primitives behind the camera, bow-tie (self-intersecting) quads, normals of length 0"""
    o = OracleScene.load(os.path.join(SCENES, "cbox_quads.obj" if quads else "cbox.obj"))
    p = o.prims()
    types = list(p["type"]); verts = [v.copy() for v in p["verts"]]; normal = list(p["normal"]); bsdf = list(p["bsdf"]); Le = list(p["Le"])

    def add(t, vs, n=(0, 0, 1), kd=(0.6, 0.5, 0.4), le=(0, 0, 0)):
        v = np.zeros((4, 3), F); v[:len(vs)] = np.asarray(vs, F)
        types.append(t); verts.append(v); normal.append(np.asarray(n, F)); bsdf.append(np.asarray(kd, F)); Le.append(np.asarray(le, F))

    c = np.array([0.0, 2.5, -2.0], F)
    add(0, [c, c, c])                                                    # a point
    add(0, [c, c + [1, 0, 0], c + [2, 0, 0]])                            # collinear
    add(0, [c, c + [1, 0, 0], c])                                        # duplicated vertex
    add(0, [c, c + [1e-20, 0, 0], c + [0, 1e-20, 0]])                    # denormal-scale determinant
    add(0, [c + [0, 0, 1], c + [1e-4, 0, 1], c + [0, 1e-4, 1]], le=(3, 3, 3))   # tiny but valid emitter
    add(0, [[-1e6, 0.5, -3], [1e6, 0.5, -3], [0, 0.5, 1e6]], n=(0, 1, 0))        # huge, axis-aligned, cuts through the box
    add(0, [[-1, 1, -1], [1, 1, -1], [0, 3, -1]], n=(0, 0, 0))                   # zero normal
    add(0, [[-1, 1, 20], [1, 1, 20], [0, 3, 20]])                                # behind the camera
    if quads:
        add(1, [c, c, c, c])                                             # degenerate quad
        add(1, [[-1, 1, -2], [1, 1, -2], [-1, 3, -2], [1, 3, -2]])       # bow-tie
        add(1, [[-0.5, 0.2, -1], [0.5, 0.2, -1], [0.5, 0.2, -2], [-0.5, 0.2, -2]], n=(0, 1, 0), le=(0.5, 0.5, 2))   # axis-aligned emitter
    wall = verts[2].copy()
    add(int(types[2]), wall[:4 if types[2] else 3], n=normal[2], kd=(0.9, 0.1, 0.1))   # exact duplicate of an existing face
    return (np.array(types, np.int32), np.stack(verts).astype(F), np.stack(normal).astype(F), np.stack(bsdf).astype(F), np.stack(Le).astype(F))


@pytest.mark.parametrize("quads", [False, True])
def test_degenerate_geometry_matches_oracle(R, quads):
    rng = np.random.default_rng(13)
    arrs = _degenerate_scene(rng, quads)
    R.load_scene_arrays(*arrs)
    o = OracleScene.from_arrays(*arrs)
    O, D = _test_rays(o, rng, 3000)
    g = R.debug_intersect(O, D)
    for i in range(len(O)):
        h = o.intersect(O[i], D[i])
        assert g["hit"][i] == h.hit and g["prim"][i] == h.prim, i
        if h.hit:
            assert bits(g["t"][i]) == bits(F(h.t))
    W, H, spp = 72, 64, 8
    R.update_resolution(W, H)
    for mode in (-1, R.LANE, R.PHASED):
        R.set_traversal(mode)
        R.update_resolution(W, H)
        R.set_config(spp=spp, max_depth=6, collect_stats=True)
        st = R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad, ost = o.render(default_camera(), W, H, spp, max_depth=6)
        assert_same_image(rgb, rad, orgb, orad, f"degenerate quads={quads} mode={mode}")
        assert (st.rays, st.node_visits, st.prim_tests, st.hits) == (ost.rays, ost.node_visits, ost.prim_tests, ost.hits)
    R.set_traversal(-1)
    R.run_radiosity_solver(mc_samples=8, num_iterations=2)
    got = R.radiosity_solution()
    exp = o.radiosity_solve(mc_samples=8, num_iterations=2)
    for k in ("form_factors", "radiosity", "grid"):
        same = (bits(got[k]) == bits(exp[k])) | (np.isnan(got[k]) & np.isnan(exp[k]))
        assert same.all(), (k, int((~same).sum()))
    R.set_config(collect_stats=False)


def test_whole_path_from_plain_c(R, tmp_path):
    """examples/render_frame.c (C99): solver, guided frame, radiosity view and PNG through the C ABI; the frame it reports
    is the one the Python binding gets for the same calls."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "cuda-pathtracer_amd")
    exe = tmp_path / "render_frame"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "render_frame.c"),
                           "-L" + libdir, "-lptmi", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    png = tmp_path / "c.png"
    out = subprocess.run([str(exe), os.path.join(SCENES, "cbox.obj"), str(png)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "radiosity: 128 primitives, 16384 pairs" in out.stdout and png.stat().st_size > 160 * 120 * 3
    R.load_scene(os.path.join(SCENES, "cbox.obj"), 1, False)
    R.run_radiosity_solver(mc_samples=16, num_iterations=4)
    R.update_resolution(160, 120); R.set_config(spp=8, max_depth=5, sampling_mode=3)
    R.render_frame()
    _, rad = R.read_image()
    mean = float(rad.astype(np.float64).sum() / rad.size)
    assert f"mean radiance {mean:.6f}" in out.stdout, (mean, out.stdout)
    R.set_config(sampling_mode=0)


def test_tiled_frame_from_plain_c_through_rccl(R, tmp_path):
    """examples/render_tiled.c (C99): ptmi_dist_unique_id / ptmi_dist_init / ptmi_gather_frame / ptmi_read_frame - the multi-GPU
    path of a C caller - with the one rank a one-GPU box allows; the frame it writes is the Python binding's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "cuda-pathtracer_amd")
    exe = tmp_path / "render_tiled"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "render_tiled.c"),
                           "-L" + libdir, "-lptmi", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    png = tmp_path / "t.png"
    out = subprocess.run([str(exe), os.path.join(SCENES, "cbox.obj"), "1", "0", str(tmp_path / "id"), str(png), "96", "64", "4"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    R.load_scene(os.path.join(SCENES, "cbox.obj"))
    R.update_resolution(96, 64); R.set_config(spp=4, max_depth=5)
    R.render_frame()
    rgb, _ = R.read_image()
    assert f"byte sum {int(rgb.astype(np.uint64).sum())}" in out.stdout, out.stdout
    assert png.stat().st_size > 96 * 64 * 3
