#!/usr/bin/env python3
"""Where do the exact walk and the fast tree name different hits?  tools/fast_mismatch.py [million rays=20]
Random rays of the kind the path tracer casts in the 1 M-triangle scene (origins on the walls, cosine-ish directions, and camera
rays), through ptmi_debug_intersect (the reference's tree) and ptmi_debug_intersect_fast; every disagreement is printed with
both hits, classified as a tie (same t, other triangle) or a miss of one of the walks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, ptmi_scenes
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20
base = ptmi.HostScene.load(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj")).prims()
sc = ptmi_scenes.tessellated_cornell(base, 256, 128)
r = ptmi.Renderer(0)
r.load_scene_arrays(sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
r.debug_set_fast_tree(3)
rng = np.random.default_rng(7)
tot = ties = other = 0
for chunk in range(M):
    n = 1_000_000
    # first hop: random rays from inside the box give hit points on the walls; second hop starts there
    o0 = rng.uniform([-2.4, 0.3, -5.0], [1.9, 4.9, -0.6], (n, 3)).astype(np.float32)
    d0 = rng.normal(size=(n, 3)).astype(np.float32); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    h0 = r.debug_intersect(o0, d0)
    ok = h0["hit"] > 0
    o1 = (h0["p"][ok] + 1e-4 * h0["n"][ok] * np.sign(-(d0[ok] * h0["n"][ok]).sum(1, keepdims=True))).astype(np.float32)
    d1 = rng.normal(size=(len(o1), 3)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    for o, d in ((o0, d0), (o1, d1)):
        e = r.debug_intersect(o, d); f = r.debug_intersect_fast(o, d)
        ep = np.where(e["hit"] > 0, e["prim"], -1); fp = np.where(f["hit"] > 0, f["prim"], -1)
        bad = np.nonzero((ep != fp) | (np.where(e["hit"] > 0, e["t"], 0).view(np.uint32) != f["t"].view(np.uint32)))[0]
        tot += len(o)
        for i in bad:
            tie = ep[i] >= 0 and fp[i] >= 0 and e["t"][i] == f["t"][i]
            ties += tie; other += not tie
            print(f"ray o={o[i]} d={d[i]}: exact prim {ep[i]} t {e['t'][i]:.9g} | fast prim {fp[i]} t {f['t'][i]:.9g} -> {'TIE (same t)' if tie else 'different t'}", flush=True)
    print(f"[{chunk + 1} M first-hop + their second-hop rays] rays {tot}, ties {ties}, other {other}", flush=True)
