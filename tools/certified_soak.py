#!/usr/bin/env python3
"""Differential soak of the automatic (certified) walk against the reference's tree ON THE GPU: random soups of triangles and skewed
quads (65 .. 20 000 primitives, some with many coplanar / duplicated primitives so that ties and flat boxes are common), random
cameras, path tracing and the radiosity pre-pass - and, from 9 scenes up, the inputs on which the walk's proof is thinnest
(hard_scene: skimmed sheets, needles, stacked layers, degenerate primitives, a far-away outlier) - every frame and every solution
must be bit-identical between the two walks.
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
def soup(n, kind):
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
    return types, verts


def finish(types, verts):
    n = len(types)
    normal = rng.normal(0, 1, (n, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    bsdf = rng.uniform(0.1, 0.95, (n, 3)).astype(F)
    Le = (rng.uniform(0, 6, (n, 3)) * (rng.random((n, 1)) < 0.1)).astype(F)
    return np.asarray(types, np.int32), np.asarray(verts, F), normal.astype(F), bsdf, Le


def tris(list_of_3x3):
    v = np.zeros((len(list_of_3x3), 4, 3), F); v[:, :3] = np.asarray(list_of_3x3, F)
    return np.zeros(len(list_of_3x3), np.int32), v


def hard_scene(kind, it):
    """The inputs on which the certified walk's argument is thinnest (the lemma's precondition): rays that skim a
    displaced sheet, needle triangles, stacked near-coplanar layers of large triangles, a sloppy exporter's degenerate primitives
    in a scene large enough for the 8-wide walks, one primitive a million units away.  Returns (scene arrays, camera, note)."""
    cam = ptmi.default_camera()
    cam.orbit = 0
    if kind == 4:                                # a displaced sheet (8 192 triangles, relief 1e-3) + a wall; the camera skims it at 1e-7 .. 1e-3 rad
        g = 64
        xs = np.linspace(-4, 4, g + 1); zs = np.linspace(-4, 4, g + 1)
        y = rng.uniform(-1e-3, 1e-3, (g + 1, g + 1))
        P = lambda i, j: (xs[i], y[i, j], zs[j])
        t = []
        for i in range(g):
            for j in range(g):
                t.append([P(i, j), P(i + 1, j), P(i + 1, j + 1)]); t.append([P(i, j), P(i + 1, j + 1), P(i, j + 1)])
        t.append([(4.5, -3, -6), (4.5, 6, -6), (4.5, 6, 6)]); t.append([(4.5, -3, -6), (4.5, 6, 6), (4.5, -3, 6)])
        tilt = [1e-7, 1e-5, 1e-3, 1e-6, 3e-7, 1e-4][it % 6]
        h = 1.5e-3 + tilt * 4.0
        cam.origin[:] = (-4.4, h, 0.37); cam.lookat[:] = (4.4, h - tilt * 8.8, 0.41); cam.vfov_deg = 0.02
        return finish(*tris(t)), cam, f"sheet skimmed at {tilt:g} rad"
    if kind == 5:                                # a fence of 4 096 needles (short edge 1e-8 .. 1e-5, pitch twice that) in front of a soup
        w = [1e-8, 1e-6, 1e-5, 1e-7][it % 4]
        k = np.arange(4096)
        x0 = (k * 2.0 * w).astype(np.float64)
        t = [[(x, 1.0, -1.0), (x + w, 1.0, -1.0), (x, 2.0, -1.0 + (1e-3 if i % 2 else 0.0))] for i, x in enumerate(x0)]
        ty, tv = tris(t)
        sy, sv = soup(700, 1)
        sv[:, :, 2] -= 2.0
        cam.origin[:] = (float(x0.mean()), 1.5, 4.0); cam.lookat[:] = (float(x0.mean()), 1.5, -1.0)
        cam.vfov_deg = float(np.degrees(2.0 * np.arctan(max(x0[-1] * 0.6, 2e-6) / 5.0)))
        return finish(np.concatenate([ty, sy]), np.concatenate([tv, sv])), cam, f"needles, short edge {w:g}"
    if kind == 6:                                # a soup of 700 + what a sloppy exporter produces (tests/test_gpu_parity.py: _degenerate_scene)
        ty, tv = soup(700, 0)
        c = np.array([0.0, 2.5, -2.0])
        d = [[c, c, c], [c, c + [1, 0, 0], c + [2, 0, 0]], [c, c + [1, 0, 0], c], [c, c + [1e-20, 0, 0], c + [0, 1e-20, 0]],
             [c + [0, 0, 1], c + [1e-4, 0, 1], c + [0, 1e-4, 1]], [[-1e6, 0.5, -3], [1e6, 0.5, -3], [0, 0.5, 1e6]],
             [[-1, 1, -1], [1, 1, -1], [0, 3, -1]], [[-1, 1, 20], [1, 1, 20], [0, 3, 20]]]
        dy, dv = tris(d)
        q = np.zeros((2, 4, 3), F); q[0] = c; q[1] = [[-1, 1, -2], [1, 1, -2], [-1, 3, -2], [1, 3, -2]]      # degenerate quad, bow-tie
        arr = finish(np.concatenate([ty, dy, np.ones(2, np.int32)]), np.concatenate([tv, dv, q]))
        arr[2][len(ty) + 6] = 0.0                # a normal of length 0
        cam.origin[:] = (rng.uniform(-1, 1), rng.uniform(1, 4), rng.uniform(2, 9))
        return arr, cam, "degenerate primitives in a soup of 700 (one face of 1e6 units)"
    if kind == 7:                                # six layers of two 12-unit triangles 1e-4 apart + a soup; rays at 1e-8 .. 1e-5 rad between the layers
        t = []
        for k in range(6):
            yk = k * 1e-4
            t.append([(-6, yk, -6), (6, yk, -6), (6, yk, 6)]); t.append([(-6, yk, -6), (6, yk, 6), (-6, yk, 6)])
        ty, tv = tris(t)
        sy, sv = soup(300, 1)
        sv[:, :, 1] += 1.0
        tilt = [1e-8, 1e-6, 1e-5, 1e-7][it % 4]
        h = 2.5e-4 + (it % 3) * 1e-4
        cam.origin[:] = (-7.0, h, 0.2); cam.lookat[:] = (7.0, h - tilt * 14.0, 0.3); cam.vfov_deg = 0.002
        return finish(np.concatenate([ty, sy]), np.concatenate([tv, sv])), cam, f"stacked layers, {tilt:g} rad"
    # kind 8: a soup of 3 000 and ONE triangle a million units away (round 3: the scene's largest coordinate set every pad and every eps)
    ty, tv = soup(3000, 1)
    oy, ov = tris([[(1e6, 1e6, -1e6), (1e6 + 10, 1e6, -1e6), (1e6, 1e6 + 10, -1e6)]])
    cam.origin[:] = (rng.uniform(-1, 1), rng.uniform(1, 4), rng.uniform(2, 9))
    return finish(np.concatenate([ty, oy]), np.concatenate([tv, ov])), cam, "one primitive a million units away"


seen = {}
kinds = [0, 1, 2, 3] if n_scenes <= 8 else [0, 1, 2, 3, 4, 5, 6, 7, 8, 4, 5, 7]
for it in range(n_scenes):
    kind = kinds[it % len(kinds)]               # 0 generic soup, 1 triangles only, 2 axis-aligned (flat boxes), 3 with duplicates (ties), 4 .. 8: hard_scene
    note = ""
    if kind <= 3:
        n = int(rng.choice([65, 130, 700, 3000, 9000, 20000]))
        arrays = finish(*soup(n, kind))
        cam = ptmi.default_camera()
        cam.origin[:] = (rng.uniform(-1, 1), rng.uniform(1, 4), rng.uniform(2, 9) if it % 5 else 40.0)     # every fifth camera far away
    else:
        seen[kind] = seen.get(kind, -1) + 1           # the k-th scene of its kind takes the k-th value of the kind's list (hardest first)
        arrays, cam, note = hard_scene(kind, seen[kind])
    types = arrays[0]; n = len(types)
    R.load_scene_arrays(*arrays)
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
    line = f"scene {it}: kind {kind}{' (' + note + ')' if note else ''}, {n} primitives ({int(types.sum())} quads), walk {frames[1][0]}: {nd} pixels differ; {st.hits} hits, chain {st.cert_chain}, reference's walk {st.cert_fallback}"
    if kind == 8:
        # one far-away primitive must not push the scene's hits off the one-fetch certificate (eps and pads follow each box's own scale)
        ok = frames[1][0] == R.CERTIFIED and st.cert_chain <= 0.05 * max(st.hits, 1) and st.cert_fallback <= 1e-3 * max(st.hits, 1)
        line += f"; chain / hits {st.cert_chain / max(st.hits, 1):.4f} (bound 0.05), reference's walk / hits {st.cert_fallback / max(st.hits, 1):.2e} (bound 1e-3): {'ok' if ok else 'OVER'}"
        bad += 0 if ok else 1
    if n <= 3000 and kind <= 3:                  # the pre-pass too
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
R.set_camera(ptmi.default_camera())
print(f"{n_scenes} scenes: {bad} mismatches; certified walk: {tot['hits']} hits, {tot['chain']} through the chain, {tot['fallback']} through the reference's walk")
sys.exit(1 if bad else 0)
