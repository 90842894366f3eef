"""The opt-in FAST tree (ptmi_config.fast_tree; cuda-pathtracer_amd/csrc/wide_bvh.h).

SURVEY 7 (last bullet): "may offer an SAH/wide tree for speed runs (results identical except exact-tie cases, which must be
reported)"; BASELINE's bar is pixel RMSE < 1e-4.  The fast walk tests the reference's triangles with the reference's
arithmetic and only chooses other boxes on the way, so these tests first ask for MORE than the bar - the same triangle and
the same t for every ray, the same frame bit for bit - report what differs (the count is printed and, for the frames,
bounded), and assert the bar itself (RMSE < 1e-4) on the rows of BASELINE configs[4] that the exact test renders.

CPU part (no GPU): the builder and the host-side walk, which takes the kernel's decisions one for one, against the oracle's
Scene::intersect.  GPU part: the kernel's walk against the host walk (same hits, same visit counts), frames against the
oracle.
"""
import os

import numpy as np
import pytest

import ptmi
import ptmi_scenes
from oracle_binding import OracleScene, SCENES, default_camera

F = np.float32
FLT_MAX = 3.4028234663852886e38


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def tess(cu, cv):
    base = ptmi.HostScene.load(os.path.join(SCENES, "cbox_quads.obj")).prims()
    sc = ptmi_scenes.tessellated_cornell(base, cu, cv, seed=1)
    return (sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])


def soup(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-2, 2, (n, 1, 3)).astype(F)
    v = np.zeros((n, 4, 3), F); v[:, :3] = c + rng.uniform(-0.3, 0.3, (n, 3, 3)).astype(F)
    nr = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); nr /= np.maximum(np.linalg.norm(nr, axis=1, keepdims=True), 1e-20)
    return (np.zeros(n, np.int32), v, nr.astype(F), rng.uniform(0.2, 0.9, (n, 3)).astype(F), (rng.uniform(0, 1, (n, 3)) > 0.9).astype(F) * 5)


def rays(n, seed):
    """half from the default camera's side into the box, half from inside it; plus axis-parallel and grazing ones"""
    rng = np.random.default_rng(seed)
    o = np.concatenate([np.tile(np.array([[0.0, 2.5, 8.5]], F), (n // 2, 1)), rng.uniform([-2.5, 0.2, -5], [2.0, 5, -0.5], (n - n // 2, 3)).astype(F)])
    d = rng.normal(size=(n, 3)).astype(F); d[: n // 2, 2] = -np.abs(d[: n // 2, 2]) - 2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n - n // 2
    ax = np.zeros((min(60, k), 3), F)
    for i in range(len(ax)):
        ax[i, i % 3] = 1.0 if (i // 3) % 2 else -1.0                     # exactly axis-parallel: 1/0 slopes
    d[n // 2: n // 2 + len(ax)] = ax
    return o, d.astype(F)


def oracle_hits(o, org, d, t_min=1e-4, t_max=FLT_MAX):
    hs = [o.intersect(org[i], d[i], t_min, t_max) for i in range(len(org))]
    hit = np.array([h.hit for h in hs], bool)
    return (np.where(hit, np.array([h.prim for h in hs]), -1), np.where(hit, np.array([h.t for h in hs], F), F(0)),
            float(np.mean([h.node_visits for h in hs])), float(np.mean([h.prim_tests for h in hs])))


SCENE_CASES = [("cbox", None), ("cbox_sub3", None), ("tess32x16", None), ("soup3000", None), ("soup2", None),
               ("quads_sub3", None), ("mixed_soup", None)]


def scene_arrays(name):
    if name == "cbox":
        p = ptmi.HostScene.load(os.path.join(SCENES, "cbox.obj")).prims()
    elif name == "cbox_sub3":
        p = ptmi.HostScene.load(os.path.join(SCENES, "cbox.obj"), 3, False).prims()
    elif name == "tess32x16":
        return tess(32, 16)
    elif name == "soup3000":
        return soup(3000, 5)
    elif name == "soup2":
        return soup(2, 6)
    elif name == "quads_sub3":                                   # native quads (Quad::intersect's two halves in a wide leaf)
        p = ptmi.HostScene.load(os.path.join(SCENES, "cbox_quads.obj"), 3, False).prims()
    elif name == "mixed_soup":                                   # triangles and quads in one scene
        t = soup(1500, 7)
        q = ptmi.HostScene.load(os.path.join(SCENES, "cbox_quads.obj"), 2, False).prims()
        return tuple(np.concatenate([a, q[k]]) for a, k in zip(t, ("type", "verts", "normal", "bsdf", "Le")))
    return (p["type"], p["verts"], p["normal"], p["bsdf"], p["Le"])


@pytest.mark.parametrize("name", [c[0] for c in SCENE_CASES])
@pytest.mark.parametrize("max_leaf", [1, 3])
def test_host_walk_finds_the_oracles_hits(name, max_leaf):
    """Same triangle and same t (bit for bit) as Scene::intersect for every ray, incl. [t_min, t_max] windows."""
    arrs = scene_arrays(name)
    hs = ptmi.HostScene.from_arrays(*arrs)
    info = hs.fast_tree_build(max_leaf)
    assert info["n_nodes"] >= 1 and 1 <= info["depth"] <= 48
    o = OracleScene.from_arrays(*arrs)
    org, d = rays(3000, 11)
    for t_min, t_max in ((1e-4, FLT_MAX), (0.5, 9.0)):
        prim, t, nodes_exact, tests_exact = oracle_hits(o, org, d, t_min, t_max)
        r = hs.fast_tree_intersect(org, d, t_min, t_max)
        bad = int((prim != r["prim"]).sum() + (bits(t) != bits(r["t"])).sum())
        print(f"{name} max_leaf {max_leaf} [{t_min}, {t_max:.3g}]: {int((prim >= 0).sum())} hits, {bad} differ; node visits per ray "
              f"{r['node_visits'] / len(org):.2f} (exact walk {nodes_exact:.2f}), triangle tests {r['prim_tests'] / len(org):.2f} ({tests_exact:.2f})")
        assert bad == 0
        assert r["max_stack"] < info["depth"]


def test_fast_tree_walk_needs_a_built_tree():
    hs = ptmi.HostScene.load(os.path.join(SCENES, "cbox_quads.obj"))
    with pytest.raises(ptmi.PtmiError):
        hs.fast_tree_intersect(np.zeros((1, 3), F), np.ones((1, 3), F))      # nothing built
    assert hs.fast_tree_build()["n_nodes"] >= 1                              # quads are fine: a quad is one primitive of a wide leaf


def test_coplanar_ties_keep_the_references_choice():
    """Rays aimed at shared edges and vertices of coplanar triangles: both are hit at the same t, the reference keeps the one it
    visits first (scene.h:89-90); the fast walk must name the same triangle (cbox.obj's left wall halves carry different vn)."""
    p = ptmi.HostScene.load(os.path.join(SCENES, "cbox.obj"), 2, False).prims()
    arrs = (p["type"], p["verts"], p["normal"], p["bsdf"], p["Le"])
    hs = ptmi.HostScene.from_arrays(*arrs); hs.fast_tree_build(3)
    o = OracleScene.from_arrays(*arrs)
    v = p["verts"][:, :3]
    targets = np.concatenate([v.reshape(-1, 3), 0.5 * (v[:, 0] + v[:, 1]), 0.5 * (v[:, 1] + v[:, 2]), 0.5 * (v[:, 0] + v[:, 2])]).astype(F)
    org = np.tile(np.array([[0.1, 2.6, 3.0]], F), (len(targets), 1))
    d = targets - org; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    prim, t, _, _ = oracle_hits(o, org, d)
    r = hs.fast_tree_intersect(org, d)
    ties = 0
    for i in range(len(org)):                       # how many of these rays really are ties: a second triangle at the same t
        if prim[i] >= 0:
            h2 = o.intersect(org[i], d[i], 1e-4, FLT_MAX, use_bvh=False)
            ties += int(h2.hit and h2.prim != prim[i] and F(h2.t) == t[i])
    bad = int((prim != r["prim"]).sum() + (bits(t) != bits(r["t"])).sum())
    print(f"{len(org)} edge/vertex rays, {bad} differ from the reference's choice")
    assert bad == 0


# ------------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def R():
    r = ptmi.Renderer(0)
    yield r
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox", "cbox_sub3", "tess32x16", "soup3000", "quads_sub3", "mixed_soup"])
def test_gpu_walk_equals_host_walk(R, name):
    """ptmi_debug_intersect_fast (the kernel's walk) against the host walk: same hits, same t, and the same number of node
    visits and triangle tests in total - the two take the same decisions - and both equal the oracle's hits."""
    arrs = scene_arrays(name)
    R.load_scene_arrays(*arrs)
    info = R.debug_set_fast_tree(3)
    hs = ptmi.HostScene.from_arrays(*arrs); hinfo = hs.fast_tree_build(3)
    assert (info["n_nodes"], info["depth"]) == (hinfo["n_nodes"], hinfo["depth"])
    org, d = rays(20000, 3)
    g = R.debug_intersect_fast(org, d)
    h = hs.fast_tree_intersect(org, d)
    gp = np.where(g["hit"] > 0, g["prim"], -1)
    assert (gp == h["prim"]).all() and (bits(g["t"]) == bits(h["t"])).all()
    assert (g["node_visits"], g["prim_tests"]) == (h["node_visits"], h["prim_tests"])
    e = R.debug_intersect(org, d)                    # the exact walk on the GPU (already checked against the oracle elsewhere)
    assert (np.where(e["hit"] > 0, e["prim"], -1) == gp).all() and (bits(np.where(e["hit"] > 0, e["t"], 0)) == bits(g["t"])).all()


def frame_diff(a, b):
    nd = int((bits(a) != bits(b)).any(axis=-1).sum())
    rmse = float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))
    return nd, rmse, float(np.abs(a.astype(np.float64) - b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name,sub,W,H,spp,depth", [("cbox.obj", 0, 160, 120, 16, 8), ("cbox.obj", 3, 128, 128, 8, 5)])
def test_fast_frames_of_small_scenes(R, name, sub, W, H, spp, depth):
    """Whole frames through ptmi_bounce_wide against the oracle: reported, and here required to be identical (no tie or grazing
    case is expected to reach a pixel of these views); every segments-per-launch setting and a frame batch give the same."""
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, False); R.update_resolution(W, H)
    R.set_config(spp=spp, max_depth=depth, fast_tree=True, segments_per_launch=0, collect_stats=True)
    st = R.render_frame()
    rgb, rad = R.read_image()
    orgb, orad, ost = OracleScene.load(path, sub, False).render(default_camera(), W, H, spp, max_depth=depth)
    nd, rmse, mx = frame_diff(rad, orad)
    print(f"{name} sub {sub} {W}x{H}x{spp}: {nd} pixels differ, max abs {mx:.3e}, RMSE {rmse:.3e}; node visits per ray "
          f"{st.node_visits / st.rays:.2f} (exact {ost.node_visits / ost.rays:.2f}), triangle tests {st.prim_tests / st.rays:.2f} ({ost.prim_tests / ost.rays:.2f})")
    assert rmse < 1e-4 and nd == 0 and (rgb == orgb).all()
    assert (st.rays, st.hits) == (ost.rays, ost.hits)
    R.set_config(collect_stats=False)
    for seg in (1, 5):
        R.update_resolution(W, H); R.set_config(segments_per_launch=seg); R.render_frame()
        assert (bits(R.read_image(rgb8=False)[1]) == bits(rad)).all(), seg
    R.update_resolution(W, H); R.set_config(segments_per_launch=0)
    R.render_frames(2)
    R.select_frame(0)
    assert (bits(R.read_image(rgb8=False)[1]) == bits(rad)).all()
    R.set_config(fast_tree=False)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,frac", [(1, 0.5), (3, 0.3)])
def test_guided_sampling_through_the_fast_tree(R, mode, frac):
    """The grid / MIS branches read their PrecomputedCDF record through the hit's load-order index, which the fast walk takes
    from its own per-triangle table: guided frames through the fast tree against the oracle (cbox subdivided, 128 triangles)."""
    path = os.path.join(SCENES, "cbox.obj")
    R.load_scene(path, 1, False)
    o = OracleScene.load(path, 1, False)
    rng = np.random.default_rng(40 + mode)
    grids = (rng.random((o.n_prims, 256, 3)) ** 3).astype(F)
    grids[rng.random(o.n_prims) < 0.2] = 0.0                     # some primitives without a grid: cosine fallback
    R.set_radiosity_grids(grids); o.set_radiosity_grids(grids); o.set_mis_fraction(frac)
    W, H, spp, depth = 64, 40, 6, 5
    R.update_resolution(W, H)
    R.set_config(spp=spp, max_depth=depth, sampling_mode=mode, mis_bsdf_fraction=frac, fast_tree=True)
    try:
        R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad, _ = o.render(default_camera(), W, H, spp, max_depth=depth, sampling_mode=mode)
        nd, rmse, mx = frame_diff(rad, orad)
        print(f"guided mode {mode} through the fast tree: {nd} pixels differ, RMSE {rmse:.3e}")
        assert nd == 0 and (rgb == orgb).all()
    finally:
        R.set_config(sampling_mode=0, mis_bsdf_fraction=0.5, fast_tree=False)
        R.set_radiosity_grids(None)


@pytest.mark.gpu
@pytest.mark.parametrize("sub", [2, 3])
def test_radiosity_solver_through_the_fast_tree(R, sub):
    """ptmi_run_radiosity_solver with config.fast_tree: the visibility walk of the form-factor kernel (form_factors.h:143-208)
    through the fast tree.  An any-hit answer can differ only for a shadow ray that grazes a box the reference's own slab test
    drops by rounding (about one in 2e8 rays: 2 of 67 M form factors at 8192 primitives); on these scenes (512 and 2048
    primitives, 2.6 M / 30.7 M shadow rays) the whole solution is required to be the oracle's, bit for bit, and is reported."""
    path = os.path.join(SCENES, "cbox.obj")
    R.load_scene(path, sub, False)
    R.set_config(fast_tree=True)
    try:
        st = R.run_radiosity_solver()
        got = R.radiosity_solution()
        exp = OracleScene.load(path, sub, False).radiosity_solve(n_threads=min(os.cpu_count() or 8, 32))
        nd = int((bits(got["form_factors"]) != bits(exp["form_factors"])).sum())
        print(f"solver through the fast tree, {got['form_factors'].shape[0]} primitives, {st.rays} shadow rays: {nd} form factors differ; form factors {st.form_factor_ms:.2f} ms")
        assert st.rays == exp["rays"] and nd == 0
        for k in ("radiosity", "unshot", "grid", "radiosity_grid"):
            assert (bits(got[k]) == bits(exp[k])).all(), k
    finally:
        R.set_config(fast_tree=False)


@pytest.mark.gpu
def test_radiosity_solver_certified_walk_is_the_references(R):
    """The solver's default visibility walk from 256 triangles up: the fast tree + a proof per blocked ray that the reference's
    any-hit walk (form_factors.h:143-208) is blocked too (radiosity.hip: certified_blocked).  The whole form-factor matrix must
    equal the one the reference's own walk gives - 2048 primitives Monte-Carlo and point-to-point (centroid rays between the
    box's walls are axis-parallel: the proof's first stage does not apply, the chain of exact slab tests decides), and with
    every blocked ray forced through the chain (3) and through the reference's walk (4)."""
    path = os.path.join(SCENES, "cbox.obj")
    R.load_scene(path, 3, False)
    try:
        for mc in (1, 0):
            R.set_solver_walk(0)
            st0 = R.run_radiosity_solver(use_monte_carlo=mc, num_iterations=2)
            want = R.radiosity_solution()
            assert st0.walk == 0 and st0.cert_chain == 0
            for walk in (-1, 3, 4):
                R.set_solver_walk(walk)
                st = R.run_radiosity_solver(use_monte_carlo=mc, num_iterations=2)
                got = R.radiosity_solution()
                print(f"certified solver walk {walk}, mc {mc}: {st.rays} rays, {st.cert_chain} chains, {st.cert_fallback} fallbacks, "
                      f"form factors {st.form_factor_ms:.2f} ms (reference's walk {st0.form_factor_ms:.2f} ms)")
                assert st.walk == 2 and st.rays == st0.rays
                for k in ("form_factors", "radiosity", "unshot", "grid", "radiosity_grid"):
                    assert (bits(got[k]) == bits(want[k])).all(), (walk, mc, k)
                if walk == 3: assert st.cert_chain > 0 and st.cert_fallback == 0
                if walk == 4: assert st.cert_fallback == st.cert_chain > 0
                if walk == -1: assert st.cert_fallback <= st.cert_chain < st.rays // 10
        R.set_solver_walk(-1, 1 << 20)
        assert R.run_radiosity_solver(num_iterations=1).walk == 0                  # below the threshold: the reference's walk
        # native quads (1024 of them): certified too, and the whole solution the reference walk's
        R.load_scene(os.path.join(SCENES, "cbox_quads.obj"), 3, False)
        R.set_solver_walk(0); st0 = R.run_radiosity_solver(num_iterations=2); want = R.radiosity_solution()
        for walk in (-1, 3, 4):
            R.set_solver_walk(walk); st = R.run_radiosity_solver(num_iterations=2); got = R.radiosity_solution()
            print(f"certified solver walk {walk}, quads: {st.rays} rays, {st.cert_chain} chains, {st.cert_fallback} fallbacks, "
                  f"form factors {st.form_factor_ms:.2f} ms (reference's walk {st0.form_factor_ms:.2f} ms)")
            assert st.walk == 2 and st0.walk == 0 and st.rays == st0.rays
            for k in ("form_factors", "radiosity", "unshot", "grid", "radiosity_grid"):
                assert (bits(got[k]) == bits(want[k])).all(), (walk, "quads", k)
    finally:
        R.set_solver_walk(-1)


def render_vs_oracle(R, o, W, H, spp, depth, cam=None):
    R.update_resolution(W, H)
    R.set_config(spp=spp, max_depth=depth, collect_stats=True, segments_per_launch=0)
    st = R.render_frame()
    rgb, rad = R.read_image()
    orgb, orad, ost = o.render(cam if cam is not None else default_camera(), W, H, spp, max_depth=depth)
    nd, rmse, mx = frame_diff(rad, orad)
    return st, ost, nd, (rgb == orgb).all()


@pytest.mark.gpu
def test_certified_walk_is_exact_and_the_default_of_large_triangle_scenes(R):
    """TRAVERSAL_CERTIFIED (kernels.hip: bounce_wide_body, CERT): the fast tree's hit + a proof per ray that the reference's walk
    returns the same one, else the reference's walk for that ray.  65 536 triangles (more than the 8 192 nodes from which the
    packed layout is built): the automatic choice must be the certified walk and the frame the oracle's, bit for bit; the share
    of hits that needed the ancestor chain and of rays that were walked again is printed."""
    arrs = tess(64, 32)
    R.load_scene_arrays(*arrs)
    assert R.set_traversal(-1) == R.CERTIFIED
    o = OracleScene.from_arrays(*arrs)
    st, ost, nd, same8 = render_vs_oracle(R, o, 200, 160, 8, 8)
    print(f"certified, 65 536 triangles: {nd} pixels differ; {st.node_visits / st.rays:.2f} node fetches + {st.prim_tests / st.rays:.2f} triangle tests per ray "
          f"(reference's walk: {ost.node_visits / ost.rays:.2f} + {ost.prim_tests / ost.rays:.2f}); of {st.hits} hits {st.cert_chain} took the ancestor chain, "
          f"{st.cert_fallback} rays the reference's walk")
    assert nd == 0 and same8 and (st.rays, st.hits) == (ost.rays, ost.hits)
    assert st.cert_chain < 0.05 * st.hits and st.cert_fallback < 0.001 * st.hits
    # the same frame through the reference's own tree (packed layout) and with the certified walk cut into single segments
    R.set_traversal(R.PACKED)
    st2, _, nd2, _ = render_vs_oracle(R, o, 200, 160, 8, 8)
    assert nd2 == 0 and (st2.node_visits, st2.prim_tests) == (ost.node_visits, ost.prim_tests)
    R.set_traversal(-1)
    R.update_resolution(200, 160); R.set_config(segments_per_launch=1, collect_stats=False); R.render_frame()
    _, orad, _ = o.render(default_camera(), 200, 160, 8, max_depth=8)
    assert (bits(R.read_image(rgb8=False)[1]) == bits(orad)).all()
    R.set_config(segments_per_launch=0)


@pytest.mark.gpu
def test_certified_walk_under_guided_sampling_frame_batches_and_tiles(R):
    """Everything above the hit query on a scene where the certified walk is the automatic choice (36 864 triangles): the grid /
    MIS sampling branches (GUIDED build of ptmi_bounce_wide<..., CERT>; records reached through the walk's own load-order
    table), a pipelined batch of frames (BATCH build) and a rank's interleaved rows of a tiled frame - each against the oracle."""
    arrs = tess(48, 24)
    R.load_scene_arrays(*arrs)
    assert R.set_traversal(-1) == R.CERTIFIED
    o = OracleScene.from_arrays(*arrs)
    W, H, spp, depth = 96, 64, 5, 6
    rng = np.random.default_rng(77)
    grids = (rng.random((o.n_prims, 256, 3)) ** 3).astype(F)
    grids[rng.random(o.n_prims) < 0.2] = 0.0
    try:
        R.set_radiosity_grids(grids); o.set_radiosity_grids(grids); o.set_mis_fraction(0.4)
        for mode in (1, 3):
            R.update_resolution(W, H)
            R.set_config(spp=spp, max_depth=depth, sampling_mode=mode, mis_bsdf_fraction=0.4, collect_stats=False)
            R.render_frame()
            rgb, rad = R.read_image()
            orgb, orad, _ = o.render(default_camera(), W, H, spp, max_depth=depth, sampling_mode=mode)
            assert (bits(rad) == bits(orad)).all() and (rgb == orgb).all(), f"guided mode {mode}"
        R.set_config(sampling_mode=0, mis_bsdf_fraction=0.5); R.set_radiosity_grids(None); o.set_radiosity_grids(None)
        # three frames as one pipelined batch
        state = np.zeros((H * W, 6), np.uint32)
        want = []
        for k in range(3):
            _, orad, _ = o.render(default_camera(), W, H, spp, max_depth=depth, rng_state=state, reset_rng=(k == 0))
            want.append(orad.copy())
        R.update_resolution(W, H)
        R.render_frames(3)
        for k in (2, 0, 1):
            R.select_frame(k)
            assert (bits(R.read_image(rgb8=False)[1]) == bits(want[k])).all(), f"batch frame {k}"
        # rank 1 of 3, interleaved 4-row blocks
        R.update_resolution(W, H, n_ranks=3, rank=1, row_block=4)
        R.render_frame()
        rows = R.local_rows()
        assert (bits(R.read_image(rgb8=False)[1]) == bits(want[0].reshape(H, W, 3)[rows])).all()
    finally:
        R.set_config(sampling_mode=0, mis_bsdf_fraction=0.5, collect_stats=False); R.set_radiosity_grids(None)
        R.update_resolution(W, H)


@pytest.mark.gpu
def test_certified_walk_sends_what_it_cannot_prove_through_the_references_walk(R):
    """The three ways out of the proof, each forced: (1) EVERY hit a tie - the scene holds every triangle twice, with another
    colour, so the reference's "first visited wins" decides every pixel; (2) ray origins outside the range the boxes are padded
    for - a camera 60 units away; (3) forced onto a small scene.  Frames must be the oracle's; the counters show the path taken."""
    p = ptmi.HostScene.load(os.path.join(SCENES, "cbox.obj"), 1, False).prims()
    n = len(p["type"])
    dup = {k: np.concatenate([p[k], p[k]]) for k in ("type", "verts", "normal", "bsdf", "Le")}
    dup["bsdf"][n:] = dup["bsdf"][n:][:, ::-1] * 0.5            # the copies are darker and colour-swapped
    arrs = (dup["type"], dup["verts"], dup["normal"], dup["bsdf"], dup["Le"])
    R.load_scene_arrays(*arrs)
    o = OracleScene.from_arrays(*arrs)
    try:
        assert R.set_traversal(R.CERTIFIED) == R.CERTIFIED
        st, ost, nd, same8 = render_vs_oracle(R, o, 96, 80, 6, 5)
        print(f"every triangle twice: {nd} pixels differ; {st.cert_fallback} of {st.hits} hits decided by the reference's walk")
        assert nd == 0 and same8 and st.cert_fallback >= 0.99 * st.hits
        R.set_traversal(-1); R.set_config(fast_tree=True)             # the uncertified fast walk: the smaller-slot rule, reported
        _, _, nd_fast, _ = render_vs_oracle(R, o, 96, 80, 6, 5)
        print(f"   the same through the fast tree without the certificate: {nd_fast} pixels differ")
        R.set_config(fast_tree=False)
        # (2) a far camera on the plain scene
        path = os.path.join(SCENES, "cbox.obj")
        R.load_scene(path, 3, False)
        o = OracleScene.load(path, 3, False)
        assert R.set_traversal(R.CERTIFIED) == R.CERTIFIED
        cam = ptmi.default_camera(); cam.origin[:] = (0.5, 3.0, 60.0)
        ocam = default_camera(); ocam.origin[:] = (0.5, 3.0, 60.0)
        R.set_camera(cam)
        st, ost, nd, same8 = render_vs_oracle(R, o, 96, 80, 6, 5, cam=ocam)
        print(f"camera 60 units away: {nd} pixels differ; {st.cert_fallback} rays through the reference's walk ({96 * 80 * 6} camera rays)")
        assert nd == 0 and same8 and st.cert_fallback >= 0.9 * 96 * 80 * 6 * 0.1
        # (3) the default camera on the same small scene
        R.set_camera(ptmi.default_camera())
        st, ost, nd, same8 = render_vs_oracle(R, o, 128, 96, 8, 8)
        print(f"cbox subdivided (2048 triangles), certified: {nd} pixels differ; chain {st.cert_chain}, reference's walk {st.cert_fallback} of {st.hits} hits")
        assert nd == 0 and same8
    finally:
        R.set_camera(ptmi.default_camera()); R.set_traversal(-1); R.set_config(fast_tree=False, collect_stats=False)


@pytest.mark.gpu
@pytest.mark.parametrize("sub", [0, 2, 4])
def test_quad_scenes_through_the_fast_tree_and_the_certified_walk(R, sub):
    """Native quads: a quad is ONE primitive of a wide leaf, tested with Quad::intersect's two-half rule (quad.h:56-121).  16
    quads keep the sweep; 256 and 4096 quads walk CERTIFIED automatically.  Frames through the certified walk, through the fast
    tree without the proof and through the reference's tree must all be the oracle's (planar scene: no tie or grazing case
    reaches a pixel here)."""
    path = os.path.join(SCENES, "cbox_quads.obj")
    R.load_scene(path, sub, False)
    n = R.scene_info()["n_prims"]
    o = OracleScene.load(path, sub, False)
    W, H, spp, depth = 96, 72, 6, 6
    _, orad, ost = o.render(default_camera(), W, H, spp, max_depth=depth)
    try:
        assert R.set_traversal(-1) == (R.CERTIFIED if n > 64 else R.SWEEP)
        for mode, fast in ((-1, False), (R.CERTIFIED, False), (-1, True), (R.PHASED, False)):
            R.set_traversal(mode); R.set_config(spp=spp, max_depth=depth, fast_tree=fast, collect_stats=True)
            R.update_resolution(W, H); st = R.render_frame()
            assert (bits(R.read_image(rgb8=False)[1]) == bits(orad)).all(), (sub, mode, fast)
            assert (st.rays, st.hits) == (ost.rays, ost.hits)
            R.set_config(collect_stats=False); R.update_resolution(W, H); R.render_frame()
            assert (bits(R.read_image(rgb8=False)[1]) == bits(orad)).all(), (sub, mode, fast, "timing build")
    finally:
        R.set_traversal(-1); R.set_config(fast_tree=False, collect_stats=False)


@pytest.mark.gpu
def test_certified_walk_soak_against_the_reference_tree_on_the_gpu():
    """tools/certified_soak.py: random soups of triangles and skewed quads (65 .. 20 000 primitives; generic, triangles only,
    axis-aligned = flat boxes everywhere, every primitive twice = every hit a tie), random and far-away cameras, path tracing and
    the radiosity pre-pass: the automatic (certified) walk and the walk over the reference's tree must agree bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "certified_soak.py"), "24", "7"], capture_output=True, text=True, timeout=900)
    print(p.stdout[-4000:])
    assert p.returncode == 0, p.stdout[-5000:] + p.stderr[-2000:]
    assert "24 scenes: 0 mismatches" in p.stdout
    # the hard inputs were really there and really walked by the certified walk
    for what in ("sheet skimmed at 1e-07 rad", "needles, short edge 1e-08", "stacked layers, 1e-08 rad", "degenerate primitives", "a million units away"):
        lines = [ln for ln in p.stdout.splitlines() if what in ln]
        assert lines and all("walk 6:" in ln for ln in lines), what


@pytest.mark.gpu
def test_a_scene_the_builder_declines_renders_and_solves_through_the_references_tree(R):
    """a scene the 8-wide builder declines (a coordinate of 2e9) loads, renders and runs the radiosity pre-pass through
    the reference's tree - the automatic choices fall back and do not ask again; a walk asked for by name reports the failure."""
    args = list(soup(300, 5))
    args[1] = args[1].copy(); args[1][7, 2] = (2.0e9, 1.0, -3.0)
    R.load_scene_arrays(*args)
    assert R.set_traversal(-1) in (R.PHASED, R.PACKED)
    o = OracleScene.from_arrays(*args)
    st, ost, nd, same_rgb = render_vs_oracle(R, o, 96, 64, 4, 5)
    assert nd == 0 and same_rgb
    R.set_solver_walk(-1, 65)
    s1 = R.run_radiosity_solver(mc_samples=4, num_iterations=1)
    s2 = R.run_radiosity_solver(mc_samples=4, num_iterations=1)            # (no second attempt at the tree: same answer)
    assert s1.walk == 0 and s2.walk == 0
    with pytest.raises(ptmi.PtmiError):
        R.set_solver_walk(2, 65); R.run_radiosity_solver(mc_samples=4, num_iterations=1)
    R.set_solver_walk(-1)
    with pytest.raises(ptmi.PtmiError):
        R.set_config(fast_tree=True, spp=1); R.update_resolution(32, 32); R.render_frame()
    R.set_config(fast_tree=False)


@pytest.mark.gpu
def test_fast_tree_config5_rows(R):
    """BASELINE configs[4] (1,048,576 triangles, 2048^2, depth 8): the rows the exact test renders (tests/test_gpu_fullsize.py),
    through the fast tree: #pixels that differ from the oracle, max abs and RMSE are reported; the bar is RMSE < 1e-4."""
    W = H = 2048; depth = 8
    args = tess(256, 128)
    R.load_scene_arrays(*args)
    o = OracleScene.from_arrays(*args)
    info = R.debug_set_fast_tree(3)
    print("fast tree:", info)
    # one GPU's share (rank 3 of 8), one row at 256 spp against the oracle
    R.set_config(spp=256, max_depth=depth, segments_per_launch=0, collect_stats=True, fast_tree=True)
    R.update_resolution(W, H, n_ranks=8, rank=3, row_block=8)
    st = R.render_frame()
    rows = R.local_rows()
    _, rad = R.read_image()
    y = int(rows[100])
    _, orad, ost = o.render(default_camera(), W, H, 256, max_depth=depth, y0=y, y1=y + 1)
    nd, rmse, mx = frame_diff(rad[100], orad[y])
    print(f"config 5, row {y} at 256 spp: {nd} of {W} pixels differ, max abs {mx:.3e}, RMSE {rmse:.3e}; "
          f"node visits per ray {st.node_visits / st.rays:.2f}, triangle tests per ray {st.prim_tests / st.rays:.2f}, "
          f"LDS-served visits {st.top_node_visits / st.node_visits:.3f}")
    assert rmse < 1e-4
    assert st.node_visits / st.rays <= 30.0                      # node fetches per ray <= 30 (exact walk: 72.5)
    # the same tile at BASELINE's FULL 2048 spp (1.07 G samples through the fast tree): one row against the oracle
    R.set_config(spp=2048, collect_stats=False)
    R.update_resolution(W, H, n_ranks=8, rank=3, row_block=8)
    R.render_frame()
    _, rad = R.read_image()
    _, orad, _ = o.render(default_camera(), W, H, 2048, max_depth=depth, y0=y, y1=y + 1)
    nd, rmse, mx = frame_diff(rad[100], orad[y])
    print(f"config 5, row {y} at the full 2048 spp: {nd} of {W} pixels differ, max abs {mx:.3e}, RMSE {rmse:.3e}")
    assert rmse < 1e-4
    # whole frame at 16 spp: rows 1000..1003 against the oracle, tile union == unsharded frame
    R.set_config(spp=16, collect_stats=False)
    R.update_resolution(W, H); R.render_frame()
    _, full = R.read_image(rgb8=False)
    _, orad, _ = o.render(default_camera(), W, H, 16, max_depth=depth, y0=1000, y1=1004)
    nd, rmse, mx = frame_diff(full[1000:1004], orad[1000:1004])
    print(f"config 5, rows 1000:1004 at 16 spp: {nd} of {4 * W} pixels differ, max abs {mx:.3e}, RMSE {rmse:.3e}")
    assert rmse < 1e-4
    R.update_resolution(W, H, n_ranks=8, rank=5, row_block=8); R.render_frame()
    assert (bits(R.read_image(rgb8=False)[1]) == bits(full[R.local_rows()])).all()
    # and against the exact walk on the GPU, whole frame: how many pixels the two trees disagree on
    R.set_config(fast_tree=False)
    R.update_resolution(W, H); R.render_frame()
    _, exact = R.read_image(rgb8=False)
    nd, rmse, mx = frame_diff(full, exact)
    print(f"config 5, whole 2048^2 frame at 16 spp, fast tree vs exact walk: {nd} of {W * H} pixels differ, max abs {mx:.3e}, RMSE {rmse:.3e}")
    assert rmse < 1e-4


@pytest.mark.gpu
def test_refill_launches_and_cost_order_do_not_change_a_frame(R):
    """scenes above 64 primitives render a frame as ONE launch whose lanes take the next queued pixel when theirs is
    through, and - from the second frame after update_resolution on - in the order of the last frame's per-pixel cost.  Three successive
    frames (the RNG streams carry over) must equal, bit for bit, the same three frames rendered with both switched off (round 3's
    launch chain), with the order alone switched off, and the oracle's; more pixels than lanes, so that lanes do take second pixels."""
    args = tess(32, 16)                                   # 16 384 triangles
    R.load_scene_arrays(*args)
    assert R.set_traversal(-1) == R.CERTIFIED
    W, H, spp, depth = 1024, 640, 3, 6                    # 655 360 pixels on 393 216 lanes
    o = OracleScene.from_arrays(*args)
    state = np.zeros((H * W, 6), np.uint32)
    want = []
    for f in range(3):
        _, orad, _ = o.render(default_camera(), W, H, spp, max_depth=depth, rng_state=state, reset_rng=(f == 0))
        want.append(orad)
    saved = {k: os.environ.get(k) for k in ("PTMI_REFILL", "PTMI_ORDER")}
    try:
        for refill, order in (("1", None), ("1", "0"), ("0", "0"), ("1", "256")):
            for k, v in (("PTMI_REFILL", refill), ("PTMI_ORDER", order)):
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
            R.set_config(spp=spp, max_depth=depth, collect_stats=False, segments_per_launch=0)
            R.update_resolution(W, H)
            for f in range(3):
                st = R.render_frame()
                _, rad = R.read_image()
                nd = int((bits(rad) != bits(want[f])).any(axis=-1).sum())
                assert nd == 0, f"refill {refill} order {order} frame {f}: {nd} pixels differ from the oracle"
                if refill == "1": assert st.bounce_launches == 1
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


@pytest.mark.gpu
def test_node_test_on_the_device_equals_the_hosts(tmp_path):
    """csrc/wide_bvh.h's wide_node_test is ONE function compiled for both sides (the host walk of the 8-wide tree is the builder's own
    check); tools/node_test_check.hip runs it on 65 536 random nodes and rays on the device and on the host and compares every field."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "node_test_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "cuda-pathtracer_amd", "csrc"),
                    "-o", exe, os.path.join(root, "tools", "node_test_check.hip")], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "mismatches: child_base 0 tri_base 0 imask 0 inner 0 tris 0" in out.stdout, out.stdout + out.stderr
