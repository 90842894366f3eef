"""The radiosity pre-pass: RadiosityState::runSolver (application_state.h:688-777), form_factors.h,
grid_filter.h.  CPU: properties of the oracle's restatement (form_factors.h cannot be compiled here; the primitive.h
pieces it uses are pinned in test_oracle_vs_ref.py).  GPU: the HIP solver against the oracle, bit for bit - except the
num_iterations == 0 radiosity grid, which the reference itself accumulates with float atomics in arbitrary order."""
import json
import os

import numpy as np
import pytest

from oracle_binding import OracleScene, SCENES, default_camera, oracle_lib

F = np.float32
CBOX = os.path.join(SCENES, "cbox.obj")
QUADS = os.path.join(SCENES, "cbox_quads.obj")


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def two_facing_squares(d=1.0, side=1.0, quads=True):
    """unit squares facing each other at distance d (analytic form factor known), as 2 quads or 4 triangles"""
    h = side / 2
    lo = np.array([[-h, 0, -h], [h, 0, -h], [h, 0, h], [-h, 0, h]], F)          # normal +y must face up: order below
    hi = lo + np.array([0, d, 0], F)
    if quads:
        types = np.array([1, 1], np.int32)
        verts = np.stack([lo[[0, 3, 2, 1]], hi[[0, 1, 2, 3]]]).astype(F)        # (v00, v10, v11, v01)
        normal = np.array([[0, 1, 0], [0, -1, 0]], F)
    else:
        types = np.zeros(4, np.int32)
        verts = np.zeros((4, 4, 3), F)
        verts[0, :3] = lo[[0, 3, 2]]; verts[1, :3] = lo[[0, 2, 1]]
        verts[2, :3] = hi[[0, 1, 2]]; verts[3, :3] = hi[[0, 2, 3]]
        normal = np.array([[0, 1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0]], F)
    n = len(types)
    bsdf = np.full((n, 3), 0.5, F); Le = np.zeros((n, 3), F); Le[0] = 1.0
    return types, verts, normal, bsdf, Le


def test_form_factor_of_facing_unit_squares():
    """parallel unit squares at distance 1: F = 0.19982 (analytic).  The reference's estimator is
    visibility * mean(cos_i) * mean(cos_j) * A / (pi * mean(r)^2) - a product of means, not the mean of the kernel - so
    it is only approximately the true form factor; and with 2 primitives there are no blockers."""
    o = OracleScene.from_arrays(*two_facing_squares())
    r = o.radiosity_solve(mc_samples=256, num_iterations=1)
    ff = r["form_factors"]
    assert ff[0, 0] == 0 and ff[1, 1] == 0
    assert abs(ff[0, 1] - 0.19982) < 0.03 and abs(ff[1, 0] - 0.19982) < 0.03
    assert r["rays"] == 2 * 256                      # approx_ff >= 0.01: the full sample count, every sample valid
    assert r["grid"].sum() == 2 * 256                # every visible sample counted once in the receiver's direction grid
    # one Jacobi step: B_1 = Le_1 + rho * F_10 * Le_0, the emitter receives nothing yet
    assert np.allclose(r["radiosity"][1], 0.5 * ff[1, 0] * 1.0) and (r["radiosity"][0] == 1.0).all()
    # point-to-point: cos = 1, A = 1, r = 1 -> 1 / pi
    r = o.radiosity_solve(use_monte_carlo=False, num_iterations=0)
    assert abs(r["form_factors"][0, 1] - 1 / np.pi) < 1e-6 and r["rays"] == 2
    assert (r["radiosity_grid"] == 0).all() and (r["grid"] == 0).all()


def test_occluder_blocks_and_source_target_are_skipped():
    t, v, nr, b, le = two_facing_squares(quads=False)
    # a big blocker halfway between the squares
    blk = np.zeros((1, 4, 3), F); blk[0, :3] = [[-5, 0.5, -5], [5, 0.5, -5], [0, 0.5, 8]]
    t2 = np.concatenate([t, [0]]).astype(np.int32); v2 = np.concatenate([v, blk])
    nr2 = np.concatenate([nr, [[0, 1, 0]]]).astype(F); b2 = np.concatenate([b, [[0.5] * 3]]).astype(F)
    le2 = np.concatenate([le, [[0] * 3]]).astype(F)
    o = OracleScene.from_arrays(t2, v2, nr2, b2, le2)
    ff = o.radiosity_solve(mc_samples=32, num_iterations=0)["form_factors"]
    assert (ff[:2, 2:4] == 0).all() and (ff[2:4, :2] == 0).all()        # lower <-> upper square: blocked
    assert ff[0, 4] == 0 and ff[2, 4] > 0 and ff[4, 2] > 0            # the blocker's normal is +y: it faces the upper square only
    o1 = OracleScene.from_arrays(t, v, nr, b, le)
    ff1 = o1.radiosity_solve(mc_samples=32, num_iterations=0)["form_factors"]
    assert (ff1[:2, 2:4] > 0).all()                                     # without it they see each other (and never block themselves)


def test_solver_properties_on_the_cornell_box():
    o = OracleScene.load(CBOX)
    n = o.n_prims
    r10 = o.radiosity_solve()
    ff = r10["form_factors"]
    assert (ff >= 0).all() and (ff <= 1).all() and (np.diag(ff) == 0).all()
    assert ff.sum(1).max() < 1.3                                        # rows of a closed scene sum to ~1 (crude estimator)
    Le = o.prims()["Le"]
    assert (r10["radiosity"] >= Le).all()                               # B = Le + reflected, reflected >= 0
    # progressive structure: radiosity after k steps = Le + sum of the first k unshot terms; unshot shrinks geometrically
    r1 = o.radiosity_solve(num_iterations=1); r2 = o.radiosity_solve(num_iterations=2)
    assert (bits(r1["form_factors"]) == bits(ff)).all()                 # same streams, same form factors
    assert (bits(r2["radiosity"]) == bits(r1["radiosity"] + r2["unshot"])).all()
    assert r2["unshot"].sum() < r1["unshot"].sum() < Le.sum()
    # reflected = min(bsdf * incident, incident): with bsdf <= 1 energy never grows
    # the count grid: integer-valued, only upper-hemisphere rows (theta < pi/2 -> rows 0..7)
    g = r10["grid"].reshape(n, 16, 16)
    assert (g == np.round(g)).all() and g[:, 8:].sum() == 0 and g.sum() > 0
    rg = r10["radiosity_grid"].reshape(n, 16, 16, 3)
    assert (rg >= 0).all() and rg[:, 8:].sum() == 0 and rg.sum() > 0
    # update_radiosity_grid distributes sum_j F_ij B_j over the direction cells: the cell sum is that total
    tot = (ff[:, :, None] * r10["radiosity"][None, :, :]).sum(1)
    assert np.allclose(rg.sum((1, 2)), tot, rtol=1e-4, atol=1e-6)
    # the solver leaves the scene ready for guided sampling and the radiosity view (ui_windows.h:185-192)
    cd = o.cdfs()
    assert cd is not None and cd[:, 529].view(np.int32).sum() == n
    _, img = o.render_radiosity(default_camera(), 32, 32, 1)
    assert img.max() > 1.0 and (img > 0).mean() > 0.4                   # walls now show their radiosity (before: emitters only)


def test_filters():
    o = OracleScene.load(CBOX)
    raw = o.radiosity_solve()["radiosity_grid"].reshape(-1, 16, 16, 3)
    gau = o.radiosity_solve(enable_filtering=True, use_bilateral=False)["radiosity_grid"].reshape(-1, 16, 16, 3)
    bil = o.radiosity_solve(enable_filtering=True, use_bilateral=True)["radiosity_grid"].reshape(-1, 16, 16, 3)
    assert (gau != raw).any() and (bil != raw).any() and (gau != bil).any()
    # normalised 5x5 kernels: smoothing, bounded by the extremes; theta does not wrap (rows 8.. only get what leaks from row 7, 6)
    assert gau.max() <= raw.max() * (1 + 1e-6) and bil.max() <= raw.max() * (1 + 1e-6) and gau.min() >= 0
    assert gau[:, 10:].sum() == 0 and gau[:, 8:10].sum() > 0
    # a huge range sigma turns the bilateral weight into the gaussian one (exp(-tiny) = 1)
    big = o.radiosity_solve(enable_filtering=True, use_bilateral=True, filter_sigma_range=1e18)["radiosity_grid"].reshape(-1, 16, 16, 3)
    assert (bits(big) == bits(gau)).all()
    # a tiny range sigma keeps only equal-luminance neighbours: exp(-huge) underflows to 0 through ptmi_expf's cut-off
    tiny = o.radiosity_solve(enable_filtering=True, use_bilateral=True, filter_sigma_range=1e-6)["radiosity_grid"]
    assert np.isfinite(tiny).all()


def test_filtered_cdfs_button():
    """"Apply Filter & Rebuild CDFs" (ui_windows.h:154-167): per-primitive pdfs sum to 1 (or stay all-zero), the CDF
    records are rebuilt from the filtered luminance, "Use Raw CDFs" restores the unfiltered records."""
    from guided_fixtures import synthetic_radiosity_grids
    o = OracleScene.load(CBOX)
    with pytest.raises(RuntimeError):
        o.apply_grid_filter()                                              # no grids yet
    o.radiosity_solve(mc_samples=16)
    raw = o.cdfs().copy()
    ff, rad = o.apply_grid_filter()
    assert np.allclose(rad.sum(1), 1, atol=1e-5) and np.allclose(ff.sum(1), 1, atol=1e-5)
    assert (rad >= 0).all() and (ff >= 0).all()
    flt = o.cdfs()
    assert (flt[:, :256] == rad).all() and (flt.view(np.uint32) != raw.view(np.uint32)).any()
    assert np.allclose(flt[:, 528], rad.reshape(-1, 16, 16)[:, :8].sum((1, 2)), rtol=1e-5)   # total weight: upper rows only
    ffg, radg = o.apply_grid_filter(use_bilateral=False, sigma_spatial=0.8)
    assert (radg != rad).any()
    # host-supplied grids, no solver: the count grid is zero and stays zero (sum <= 1e-12: not normalised)
    o2 = OracleScene.load(CBOX)
    grids = synthetic_radiosity_grids(o2.n_prims)
    o2.set_radiosity_grids(grids)
    raw2 = o2.cdfs().copy()
    ff2, rad2 = o2.apply_grid_filter()
    assert (ff2 == 0).all()
    empty = grids.reshape(o2.n_prims, -1).sum(1) == 0
    assert empty.any() and (rad2[empty] == 0).all() and np.allclose(rad2[~empty].sum(1), 1, atol=1e-5)
    o2.set_radiosity_grids(grids)                                          # "Use Raw CDFs"
    assert (o2.cdfs().view(np.uint32) == raw2.view(np.uint32)).all()


# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def R():
    import ptmi
    r = ptmi.Renderer(0)
    yield r
    r.close()


def compare_solution(got, exp, what, exact_grid=True):
    for k in ("form_factors", "radiosity", "unshot", "grid"):
        nd = int((bits(got[k]) != bits(exp[k])).sum())
        assert nd == 0, f"{what}: {k} differs in {nd} entries (max abs {np.abs(got[k] - exp[k]).max():.3e})"
    if exact_grid:
        nd = int((bits(got["radiosity_grid"]) != bits(exp["radiosity_grid"])).sum())
        assert nd == 0, f"{what}: radiosity_grid differs in {nd} entries"
    else:
        assert np.allclose(got["radiosity_grid"], exp["radiosity_grid"], rtol=2e-5, atol=1e-7), what


@pytest.mark.gpu
@pytest.mark.parametrize("name,sub,conv,params", [
    ("cbox.obj", 0, False, {}),                                                   # the defaults: MC 64 samples, 10 steps
    ("cbox_quads.obj", 0, False, {}),                                             # quads: split sampling, strict t < max
    ("cbox.obj", 1, False, dict(mc_samples=16, num_iterations=3)),                # 128 primitives
    ("cbox_quads.obj", 1, True, dict(mc_samples=7, num_iterations=1)),            # odd sample count: n/4 -> 1, n/2 -> 3
    ("cbox.obj", 0, False, dict(use_monte_carlo=False)),                          # point-to-point
    ("cbox_quads.obj", 1, False, dict(use_monte_carlo=False, num_iterations=2)),
    ("cbox.obj", 1, False, dict(mc_samples=8, enable_filtering=True, use_bilateral=True)),
    ("cbox.obj", 0, False, dict(mc_samples=8, enable_filtering=True, use_bilateral=False, filter_sigma_spatial=0.7)),
    ("cbox.obj", 0, False, dict(mc_samples=8, enable_filtering=True, filter_sigma_range=0.05)),
])
def test_gpu_solver_matches_oracle(R, name, sub, conv, params):
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    st = R.run_radiosity_solver(**params)
    got = R.radiosity_solution()
    o = OracleScene.load(path, sub, conv)
    exp = o.radiosity_solve(**params)
    compare_solution(got, exp, f"{name} sub{sub} {params}")
    assert st.rays == exp["rays"] and st.pairs == o.n_prims ** 2
    # what the solver leaves behind: CDF records and the per-primitive radiosity of the radiosity view
    assert (R.precomputed_cdfs().view(np.uint32) == o.cdfs().view(np.uint32)).all()


@pytest.mark.gpu
def test_gpu_solver_zero_iterations(R):
    """num_iterations = 0 keeps the Monte-Carlo kernel's own radiosity grid, which the reference accumulates with float
    atomics across pairs in arbitrary order: compared to the oracle's ascending-j sums within 2e-5 relative.  Everything
    else stays exact."""
    R.load_scene(CBOX, 1, False)
    R.run_radiosity_solver(num_iterations=0, mc_samples=16)
    got = R.radiosity_solution()
    o = OracleScene.load(CBOX, 1, False)
    exp = o.radiosity_solve(num_iterations=0, mc_samples=16)
    compare_solution(got, exp, "0 iterations", exact_grid=False)
    assert (got["radiosity"] == o.prims()["Le"]).all() and exp["radiosity_grid"].sum() > 0


@pytest.mark.gpu
def test_gpu_solver_feeds_guided_sampling_and_the_radiosity_view(R):
    """end to end, as the UI does it: load, solve, render with MIS guiding; switch to the radiosity view"""
    path = CBOX
    R.load_scene(path, 1, False)
    R.run_radiosity_solver(mc_samples=16, num_iterations=4)
    o = OracleScene.load(path, 1, False)
    o.radiosity_solve(mc_samples=16, num_iterations=4)
    W = H = 48
    R.update_resolution(W, H)
    R.set_config(spp=4, max_depth=5, sampling_mode=3)
    R.render_frame()
    rgb, rad = R.read_image()
    state = np.zeros((H * W, 6), np.uint32)
    orgb, orad, _ = o.render(default_camera(), W, H, 4, sampling_mode=3, rng_state=state)
    assert (bits(rad) == bits(orad)).all() and (rgb == orgb).all()
    R.set_config(spp=2, integrator=1)
    R.render_frame()
    rgb, rad = R.read_image()
    orgb, orad = o.render_radiosity(default_camera(), W, H, 2, rng_state=state, reset_rng=False)
    assert (bits(rad) == bits(orad)).all() and (rgb == orgb).all()
    R.set_config(integrator=0, sampling_mode=0)


@pytest.mark.gpu
def test_gpu_solver_deep_tree_and_errors(R):
    """a tree deeper than 31 levels takes the explicit-stack visibility walk with the reference's drop rule (stack_ptr >= 30)"""
    import ptmi
    n = 60
    x = (2.0 ** 40 * 2.2 ** (-np.arange(n, dtype=np.float64))).astype(F)
    verts = np.zeros((n, 4, 3), F)
    for i in range(n):
        z = F(i) * F(0.01)
        verts[i, 0] = [x[i], -0.004, z]; verts[i, 1] = [x[i], 0.004, z]; verts[i, 2] = [x[i], 0.0, z + F(0.008)]
    types = np.zeros(n, np.int32)
    nr = np.tile(np.array([[1, 0, 0]], F), (n, 1)); nr[::2] = [-1, 0, 0]
    b = np.full((n, 3), 0.5, F); le = np.ones((n, 3), F)
    R.load_scene_arrays(types, verts, nr, b, le)
    assert R.scene_info()["bvh_depth"] > 31
    R.run_radiosity_solver(mc_samples=8, num_iterations=2)
    got = R.radiosity_solution()
    o = OracleScene.from_arrays(types, verts, nr, b, le)
    exp = o.radiosity_solve(mc_samples=8, num_iterations=2)
    compare_solution(got, exp, "deep tree")
    assert exp["form_factors"].sum() > 0
    with pytest.raises(ptmi.PtmiError):
        R.run_radiosity_solver(mc_samples=0)
    with pytest.raises(ptmi.PtmiError):
        R.run_radiosity_solver(num_iterations=-1)
    R.load_scene(CBOX)
    with pytest.raises(ptmi.PtmiError):
        R.radiosity_solution()                       # a scene load drops the previous solution


@pytest.mark.gpu
@pytest.mark.parametrize("bilateral,ss,sr", [(True, 1.5, 0.3), (False, 1.5, 0.3), (True, 0.6, 0.05), (False, 3.0, 1.0)])
def test_gpu_filtered_cdfs_match_oracle(R, bilateral, ss, sr):
    import ptmi
    from guided_fixtures import synthetic_radiosity_grids
    path = QUADS
    R.load_scene(path, 1, False)
    with pytest.raises(ptmi.PtmiError):
        R.apply_grid_filter()                                              # no radiosity grids yet
    R.run_radiosity_solver(mc_samples=8, num_iterations=3)
    o = OracleScene.load(path, 1, False)
    o.radiosity_solve(mc_samples=8, num_iterations=3)
    raw = R.precomputed_cdfs().copy()
    ff, rad = R.apply_grid_filter(bilateral, ss, sr)
    off, orad = o.apply_grid_filter(bilateral, ss, sr)
    assert (bits(ff) == bits(off)).all() and (bits(rad) == bits(orad)).all()
    assert (R.precomputed_cdfs().view(np.uint32) == o.cdfs().view(np.uint32)).all()
    # guided rendering uses the filtered records
    W = H = 40
    R.update_resolution(W, H); R.set_config(spp=3, sampling_mode=2)
    R.render_frame()
    rgb, radi = R.read_image()
    orgb, oradi, _ = o.render(default_camera(), W, H, 3, sampling_mode=2)
    assert (bits(radi) == bits(oradi)).all() and (rgb == orgb).all()
    R.use_raw_cdfs()
    assert (R.precomputed_cdfs().view(np.uint32) == raw.view(np.uint32)).all()
    # host-supplied grids (no solver run on this scene): count grids are zero
    R.load_scene(CBOX)
    grids = synthetic_radiosity_grids(R.scene_info()["n_prims"])
    R.set_radiosity_grids(grids)
    o2 = OracleScene.load(CBOX); o2.set_radiosity_grids(grids)
    ff, rad = R.apply_grid_filter(bilateral, ss, sr)
    off, orad = o2.apply_grid_filter(bilateral, ss, sr)
    assert (ff == 0).all() and (bits(rad) == bits(orad)).all()
    assert (R.precomputed_cdfs().view(np.uint32) == o2.cdfs().view(np.uint32)).all()
    R.set_config(sampling_mode=0)


# ------------------------------------------------------------------------------------------------------------------
# committed fixtures: tests/golden/solver_*.npz (tests/golden/make_golden.py)
# ------------------------------------------------------------------------------------------------------------------
def _solver_fixtures():
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solver_*.npz")))
    assert len(files) >= 3
    return files


def test_oracle_reproduces_committed_solver_fixtures():
    for f in _solver_fixtures():
        g = np.load(f)
        prm = json.loads(str(g["params"]))
        o = OracleScene.load(os.path.join(SCENES, str(g["scene"])), int(g["subdivision"]), False)
        sol = o.radiosity_solve(**prm)
        for k in ("form_factors", "radiosity", "unshot", "grid", "radiosity_grid"):
            assert (bits(sol[k]) == bits(g[k])).all(), (os.path.basename(f), k)
        assert sol["rays"] == int(g["rays"]) and (o.cdfs().view(np.uint32) == g["cdfs"].view(np.uint32)).all()


@pytest.mark.gpu
def test_gpu_reproduces_committed_solver_fixtures(R):
    """needs no oracle library: solver outputs, the CDF records, a MIS-guided frame and a radiosity view from the files"""
    for f in _solver_fixtures():
        g = np.load(f)
        prm = json.loads(str(g["params"]))
        R.load_scene(os.path.join(SCENES, str(g["scene"])), int(g["subdivision"]), False)
        st = R.run_radiosity_solver(**prm)
        sol = R.radiosity_solution()
        for k in ("form_factors", "radiosity", "unshot", "grid", "radiosity_grid"):
            assert (bits(sol[k]) == bits(g[k])).all(), (os.path.basename(f), k)
        assert st.rays == int(g["rays"]) and (R.precomputed_cdfs().view(np.uint32) == g["cdfs"].view(np.uint32)).all()
        W, H = int(g["width"]), int(g["height"])
        R.update_resolution(W, H)
        R.set_config(spp=4, max_depth=5, sampling_mode=3, integrator=0)
        R.render_frame()
        rgb, rad = R.read_image()
        assert (bits(rad) == bits(g["guided_radiance"])).all() and (rgb == g["guided_rgb8"]).all(), os.path.basename(f)
        R.set_config(spp=2, integrator=1)
        R.render_frame()
        rgb, rad = R.read_image()
        assert (bits(rad) == bits(g["view_radiance"])).all() and (rgb == g["view_rgb8"]).all(), os.path.basename(f)
        R.set_config(integrator=0, sampling_mode=0)
