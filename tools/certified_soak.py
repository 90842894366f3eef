#!/usr/bin/env python3
"""Differential soak of the automatic (certified) walk against the reference's tree ON THE GPU: random soups of triangles and skewed
quads (65 .. 20 000 primitives, some with many coplanar / duplicated primitives so that ties and flat boxes are common), random
cameras, path tracing and the radiosity pre-pass - every frame and every solution must be bit-identical between the two walks.
   tools/certified_soak.py [n_scenes=40] [seed=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi
F = np.float32
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
R = ptmi.Renderer(0)
bad = 0
tot = dict(hits=0, chain=0, fallback=0)
for it in range(n_scenes):
    n = int(rng.choice([65, 130, 700, 3000, 9000, 20000]))
    kind = it % 4                                # 0 generic soup, 1 triangles only, 2 axis-aligned (flat boxes), 3 with duplicates (ties)
    types = (rng.random(n) < (0.0 if kind == 1 else 0.35)).astype(np.int32)
    centers = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.2, 5.0, n), rng.uniform(-5.5, 0.5, n)], 1)[:, None, :]
    verts = (centers + rng.normal(0, 0.6 if n < 1000 else 0.25, (n, 4, 3))).astype(F)
    if kind == 2:                                # squash every primitive into an axis-aligned plane, make quads planar rectangles
        ax = rng.integers(0, 3, n)
        for a in range(3):
            m = ax == a
            verts[m, :, a] = verts[m, :1, a]
            b, c = (a + 1) % 3, (a + 2) % 3
            verts[m, 1, b] = verts[m, 0, b] + 0.4; verts[m, 1, c] = verts[m, 0, c]
            verts[m, 2, b] = verts[m, 0, b] + 0.4; verts[m, 2, c] = verts[m, 0, c] + 0.3
            verts[m, 3, b] = verts[m, 0, b]; verts[m, 3, c] = verts[m, 0, c] + 0.3
    if kind == 3:                                # the second half repeats the first: every hit there is a tie
        h = n // 2
        verts[h:2 * h] = verts[:h]; types[h:2 * h] = types[:h]
    normal = rng.normal(0, 1, (n, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    bsdf = rng.uniform(0.1, 0.95, (n, 3)).astype(F)
    Le = (rng.uniform(0, 6, (n, 3)) * (rng.random((n, 1)) < 0.1)).astype(F)
    R.load_scene_arrays(types, verts, normal.astype(F), bsdf, Le)
    cam = ptmi.default_camera()
    cam.origin[:] = (rng.uniform(-1, 1), rng.uniform(1, 4), rng.uniform(2, 9) if it % 5 else 40.0)     # every fifth camera far away
    R.set_camera(cam)
    W, H, spp = 160, 120, 6
    frames = []
    for mode in (R.PHASED, -1):
        got = R.set_traversal(mode)
        R.set_config(spp=spp, max_depth=6, collect_stats=True); R.update_resolution(W, H); st = R.render_frame()
        frames.append((got, R.read_image(rgb8=False)[1].copy(), st))
    nd = int((frames[0][1].view(np.uint32) != frames[1][1].view(np.uint32)).any(axis=-1).sum())
    st = frames[1][2]
    tot["hits"] += st.hits; tot["chain"] += st.cert_chain; tot["fallback"] += st.cert_fallback
    line = f"scene {it}: kind {kind}, {n} primitives ({int(types.sum())} quads), walk {frames[1][0]}: {nd} pixels differ; {st.hits} hits, chain {st.cert_chain}, reference's walk {st.cert_fallback}"
    if n <= 3000:                                # the pre-pass too
        sols = []
        for walk in (0, -1):
            R.set_solver_walk(walk, 65); s2 = R.run_radiosity_solver(mc_samples=8, num_iterations=2); sols.append((s2, R.radiosity_solution()))
        same = all((sols[0][1][k].view(np.uint32) == sols[1][1][k].view(np.uint32)).all() for k in ("form_factors", "radiosity", "unshot", "grid", "radiosity_grid"))
        line += f"; solver walk {sols[1][0].walk}: {'identical' if same else 'DIFFERENT'} ({sols[1][0].cert_chain} chains, {sols[1][0].cert_fallback} fallbacks of {sols[1][0].rays} rays)"
        bad += 0 if same else 1
        R.set_solver_walk(-1)
    print(line, flush=True)
    bad += 1 if nd else 0
    R.set_traversal(-1)
print(f"{n_scenes} scenes: {bad} mismatches; certified walk: {tot['hits']} hits, {tot['chain']} through the chain, {tot['fallback']} through the reference's walk")
sys.exit(1 if bad else 0)
