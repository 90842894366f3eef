#!/usr/bin/env python3
"""Exact walk vs the opt-in fast tree on the 1 M-triangle scene: tools/fast_probe.py [spp=64] [variants] [rounds=2] [shares=8,1]
variant = fast:max_leaf:c_trav:top_nodes:segments   (fast 0 = exact walk; the other fields then do not matter)
share   = n -> rank 3 of n interleaved 8-row blocks of the 2048^2 frame (8 = c5tile, 1 = c5frame)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, ptmi_scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0:3:1:80:0", "1:3:1:80:0"]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
shares = [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["8", "1"])]
base = ptmi.HostScene.load(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj")).prims()
sc = ptmi_scenes.tessellated_cornell(base, 256, 128)
r = ptmi.Renderer(0)
r.load_scene_arrays(sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
print(r.scene_info(), flush=True)
for share in shares:
    ref = None
    for v in variants:
        fast, leaf, ctrav, top, seg = v.split(":")
        info = r.debug_set_fast_tree(int(leaf), float(ctrav), 1.0, int(top)) if int(fast) else {}
        r.set_traversal(6 if int(fast) == 2 else -1)          # fast 2 = the certified (exact) form of the fast walk
        r.set_config(spp=spp, max_depth=8, segments_per_launch=int(seg), collect_stats=True, fast_tree=int(fast) == 1)
        r.update_resolution(2048, 2048, n_ranks=share, rank=min(3, share - 1), row_block=8)
        st = r.render_frame()
        rad = r.read_image(rgb8=False)[1]
        if ref is None: ref = rad
        nd = int((rad.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
        rmse = float(np.sqrt(np.mean((rad.astype(np.float64) - ref) ** 2)))
        r.set_config(collect_stats=False)
        best = 1e9
        for _ in range(rounds):
            r.update_resolution(2048, 2048, n_ranks=share, rank=min(3, share - 1), row_block=8)
            t0 = time.perf_counter(); s2 = r.render_frame(); best = min(best, time.perf_counter() - t0)
        n = 2048 * (2048 // share) * spp
        print(f"1/{share} {v:>16}: {best*1e3:8.2f} ms = {n/best/1e6:7.1f} Msamples/s, {s2.bounce_launches} launches; nodes/ray {st.node_visits/st.rays:.2f} "
              f"(LDS {st.top_node_visits/max(st.node_visits,1):.2f}) tests/ray {st.prim_tests/st.rays:.2f}; vs first variant: {nd} px differ, RMSE {rmse:.2e}; cert chain {st.cert_chain} fallback {st.cert_fallback} of {st.hits} hits; {info}", flush=True)
