"""render_radiosity (integrator.h:460-504), the "Radiosity" integrator of renderFrame (application.h:193-197): first
hit only, Le + per-primitive radiosity averaged over spp, sqrt gamma.  CPU: the oracle's restatement against values
that follow from the source by hand.  GPU: the HIP kernel against the oracle, bit for bit."""
import os

import numpy as np
import pytest

from oracle_binding import OracleScene, SCENES, default_camera

F = np.float32
CBOX = os.path.join(SCENES, "cbox.obj")


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def flat_radiosity(o, value):
    return np.tile(np.asarray(value, F)[None, :], (o.n_prims, 1))


def test_oracle_radiosity_view_known_answers():
    o = OracleScene.load(CBOX)
    W = H = 64
    # 1. zero radiosity (state after loadScene): only emitters are visible, Le = (25, 25, 25) clamps to 1 -> 255
    rgb, rad = o.render_radiosity(default_camera(), W, H, 4)
    lit = (rad > 0).any(axis=-1)
    assert 0 < lit.sum() < W * H // 8
    on = (rad == F(25.0)).all(axis=-1)                                                    # all 4 rays on the light
    assert on.any() and (rgb[on] == 255).all() and (rgb[~lit] == 0).all()
    # 2. constant radiosity c on every primitive: every pixel that sees the scene shows exactly c (4 equal terms
    #    summed and divided by 4 are exact), the byte is (uchar)(255.99f * sqrtf(c))
    c = np.array([0.25, 0.5, 0.0625], F)
    o.set_radiosity(flat_radiosity(o, c))
    rgb, rad = o.render_radiosity(default_camera(), W, H, 4)
    centre = rad[H // 2, W // 2]
    assert (centre == c).all()
    assert (rgb[H // 2, W // 2] == (F(255.99) * np.sqrt(c)).astype(np.uint8)).all()
    assert (rad[on] == (F(25.0) + c)).all()                              # color += Le; color += radiosity
    # 3. the RNG advances by exactly 2 draws per sample and is written back (integrator.h:468, 503)
    st_a = np.zeros((H * W, 6), np.uint32); st_b = np.zeros((H * W, 6), np.uint32)
    o.render_radiosity(default_camera(), W, H, 2, rng_state=st_a)
    o.render_radiosity(default_camera(), W, H, 2, rng_state=st_a, reset_rng=False)
    o.render_radiosity(default_camera(), W, H, 4, rng_state=st_b)
    assert (st_a == st_b).all()
    # ... and the path tracer continues from that state (one rand_state array serves both kernels)
    _, r1, _ = o.render(default_camera(), W, H, 2, rng_state=st_a, reset_rng=False)
    _, r2, _ = o.render(default_camera(), W, H, 2)
    assert (bits(r1) != bits(r2)).any()


def test_oracle_radiosity_is_per_primitive():
    o = OracleScene.load(CBOX)
    rng = np.random.default_rng(5)
    rad_in = rng.random((o.n_prims, 3)).astype(F)
    o.set_radiosity(rad_in)
    W = H = 32
    _, rad = o.render_radiosity(default_camera(), W, H, 1)
    # with 1 spp each pixel is exactly one primitive's Le + radiosity
    Le = o.prims()["Le"]
    table = (Le + rad_in).astype(F)
    px = rad.reshape(-1, 3)
    seen = (px > 0).any(axis=1)
    match = (px[seen][:, None, :] == table[None, :, :]).all(axis=2).any(axis=1)
    assert match.all()


# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def R():
    import ptmi
    r = ptmi.Renderer(0)
    yield r
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,sub,conv,W,H,spp", [
    ("cbox.obj", 0, False, 96, 64, 4),
    ("cbox_quads.obj", 0, False, 64, 64, 3),       # quad primitives
    ("cbox.obj", 2, False, 61, 37, 2),             # subdivided (544 prims, PHASED scene class), ragged size
    ("cbox_quads.obj", 1, True, 40, 40, 1),
])
def test_gpu_radiosity_view_matches_oracle(R, name, sub, conv, W, H, spp):
    path = os.path.join(SCENES, name)
    R.load_scene(path, sub, conv)
    R.update_resolution(W, H)
    R.set_config(spp=spp, integrator=1)
    o = OracleScene.load(path, sub, conv)
    state = np.zeros((H * W, 6), np.uint32)
    rng = np.random.default_rng(11)
    for frame, radiosity in enumerate([None, rng.random((o.n_prims, 3)).astype(F) * F(1.5)]):
        R.set_radiosity(radiosity); o.set_radiosity(radiosity)
        st = R.render_frame()
        rgb, rad = R.read_image()
        orgb, orad = o.render_radiosity(default_camera(), W, H, spp, rng_state=state, reset_rng=(frame == 0))
        assert (bits(rad) == bits(orad)).all(), f"frame {frame}: {(bits(rad) != bits(orad)).any(axis=-1).sum()} pixels differ"
        assert (rgb == orgb).all()
        assert st.samples == W * H * spp
    # switching back: the path tracer picks the shared RNG state up where the radiosity view left it
    R.set_config(spp=spp, integrator=0)
    R.render_frame()
    rgb, rad = R.read_image()
    orgb, orad, _ = o.render(default_camera(), W, H, spp, rng_state=state, reset_rng=False)
    assert (bits(rad) == bits(orad)).all() and (rgb == orgb).all()
    R.set_radiosity(None)


@pytest.mark.gpu
def test_gpu_radiosity_view_tiles_and_errors(R):
    import ptmi
    R.load_scene(CBOX)
    o = OracleScene.load(CBOX)
    rad_in = np.random.default_rng(3).random((o.n_prims, 3)).astype(F)
    o.set_radiosity(rad_in)
    W, H, spp = 64, 50, 2
    orgb, orad = o.render_radiosity(default_camera(), W, H, spp)
    for rank in range(3):                                    # interleaved row blocks, RNG keyed by the global pixel
        R.update_resolution(W, H, n_ranks=3, rank=rank, row_block=8)
        R.set_radiosity(rad_in); R.set_config(spp=spp, integrator=1)
        R.render_frame()
        rgb, rad = R.read_image()
        rows = R.local_rows()
        assert (bits(rad) == bits(orad[rows])).all() and (rgb == orgb[rows]).all()
    with pytest.raises(ptmi.PtmiError):
        R.set_radiosity(rad_in[:-1])                         # wrong primitive count
    with pytest.raises(ptmi.PtmiError):
        R.set_config(integrator=2)
    R.set_config(integrator=0); R.set_radiosity(None)
    R.update_resolution(W, H)


@pytest.mark.gpu
def test_gpu_radiosity_view_through_the_certified_walk_on_the_1m_triangle_scene(R):
    """Scenes above the sweep's 64 primitives trace the view's first hits through the certified walk: on the
    1 048 576-triangle scene two bands of rows against the oracle, and the whole 1024^2 frame against the walk over the reference's
    own tree on the GPU - bit for bit, per-primitive radiosities included (they are indexed by the reference's leaf-order slot)."""
    import time
    import ptmi_scenes
    import ptmi
    base = ptmi.HostScene.load(os.path.join(SCENES, "cbox_quads.obj")).prims()
    sc = ptmi_scenes.tessellated_cornell(base, 256, 128, seed=1)
    args = (sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
    R.load_scene_arrays(*args)
    assert R.set_traversal(-1) == R.CERTIFIED
    o = OracleScene.from_arrays(*args)
    W = H = 1024; spp = 3
    rad_in = np.random.default_rng(2).random((o.n_prims, 3)).astype(F)
    R.set_radiosity(rad_in); o.set_radiosity(rad_in)
    R.update_resolution(W, H); R.set_config(spp=spp, integrator=1)
    t0 = time.perf_counter(); R.render_frame(); t_cert = time.perf_counter() - t0
    rgb, rad = R.read_image()
    R.set_traversal(R.PACKED)                                   # the reference's tree, node for node
    R.update_resolution(W, H)
    t0 = time.perf_counter(); R.render_frame(); t_ref = time.perf_counter() - t0
    rgb2, rad2 = R.read_image()
    R.set_traversal(-1)
    assert (bits(rad) == bits(rad2)).all() and (rgb == rgb2).all(), f"{(bits(rad) != bits(rad2)).any(axis=-1).sum()} pixels differ between the two walks"
    for y0 in (300, 700):
        orgb, orad = o.render_radiosity(default_camera(), W, H, spp, y0=y0, y1=y0 + 4)
        assert (bits(rad[y0:y0 + 4]) == bits(orad[y0:y0 + 4])).all() and (rgb[y0:y0 + 4] == orgb[y0:y0 + 4]).all()
    print(f"radiosity view, 1 M triangles, 1024^2 x {spp} spp: certified walk {t_cert * 1e3:.1f} ms, reference's tree {t_ref * 1e3:.1f} ms")
    R.set_config(integrator=0); R.set_radiosity(None)
