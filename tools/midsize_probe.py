#!/usr/bin/env python3
"""Mid-size triangle scenes (between the sweep's 64 primitives and the 8192 BVH nodes from which the packed layout is built): the
phased walk over the reference's tree (rounds 1-2's walk of these scenes, forced: mode 3) against the automatic choice (the
certified walk), 1024^2 x 16 spp x depth 8.   tools/midsize_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, ptmi_scenes
r = ptmi.Renderer(0)
base = ptmi.HostScene.load(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj")).prims()
W = H = 1024; spp = 16
def run(label):
    ref = None
    for mode in (3, -1):
        got = r.set_traversal(mode)
        r.set_config(spp=spp, max_depth=8, collect_stats=True)
        r.update_resolution(W, H); st = r.render_frame()
        rad = r.read_image(rgb8=False)[1]
        if ref is None: ref = rad
        nd = int((rad.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
        r.set_config(collect_stats=False)
        best = 1e9
        for _ in range(3):
            r.update_resolution(W, H)
            t0 = time.perf_counter(); r.render_frame(); best = min(best, time.perf_counter() - t0)
        print(f"{label}: mode {mode}->{got}: {W*H*spp/best/1e6:8.1f} Msamples/s; nodes/ray {st.node_visits/st.rays:.2f} tests/ray {st.prim_tests/st.rays:.2f}; "
              f"chain {st.cert_chain/max(st.hits,1):.4f} fallback {st.cert_fallback/max(st.hits,1):.6f} of hits; {nd} px differ", flush=True)
    r.set_traversal(-1)
for cu, cv in ((4, 2), (8, 4), (16, 8), (22, 11)):
    sc = ptmi_scenes.tessellated_cornell(base, cu, cv)
    r.load_scene_arrays(sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
    run(f"tess {cu}x{cv} ({r.scene_info()['n_prims']} triangles, {r.scene_info()['n_bvh_nodes']} nodes)")
for sub in (1, 2, 3, 4):
    r.load_scene(os.path.join(ROOT, "tests/golden/scenes/cbox.obj"), sub, False)
    run(f"cbox sub {sub} ({r.scene_info()['n_prims']} triangles, {r.scene_info()['n_bvh_nodes']} nodes)")
