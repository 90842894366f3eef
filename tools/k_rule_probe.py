#!/usr/bin/env python3
"""Whole-frame-per-launch rule (segments_per_launch = 0) against 32 segments per launch on mid-size scenes (PHASED walk)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi
r = ptmi.Renderer(0)
scene = os.path.join(ROOT, "tests", "golden", "scenes", "cbox.obj")
for sub in (2, 3, 4):
    r.load_scene(scene, sub, False)
    info = r.scene_info()
    for side in (256, 512, 724, 1024):
        res = {}
        for seg in (32, 0):
            r.set_config(spp=64, max_depth=8, segments_per_launch=seg, collect_stats=False)
            r.update_resolution(side, side)
            r.render_frame()
            rad = r.read_image(rgb8=False)[1]
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); st = r.render_frame(); ts.append(time.perf_counter() - t0)
            res[seg] = (min(ts), st.bounce_launches, rad)
        r.update_resolution(side, side); r.set_config(segments_per_launch=32); r.render_frame(); a = r.read_image(rgb8=False)[1]
        r.update_resolution(side, side); r.set_config(segments_per_launch=0); r.render_frame(); b = r.read_image(rgb8=False)[1]
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
        print(f"sub {sub} ({info['n_prims']} prims, mode {r.set_traversal(-1)}) {side}^2: K=32 {res[32][0]*1e3:7.2f} ms ({res[32][1]} launches)  auto {res[0][0]*1e3:7.2f} ms ({res[0][1]} launches)  ratio {res[32][0]/res[0][0]:.3f}", flush=True)
