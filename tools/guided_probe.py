#!/usr/bin/env python3
"""Timing of the guided sampling modes on mid-size scenes walked by ptmi_bounce_phased (cbox subdivided, grids from the solver)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi
r = ptmi.Renderer(0)
scene = os.path.join(ROOT, "tests", "golden", "scenes", "cbox.obj")
for sub in (3, 4):
    r.load_scene(scene, sub, False)
    r.run_radiosity_solver()
    for side in (512, 1024):
        for mode in (0, 2, 3):
            r.set_config(spp=64, max_depth=8, sampling_mode=mode, collect_stats=False)
            r.update_resolution(side, side); r.render_frame()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); st = r.render_frame(); ts.append(time.perf_counter() - t0)
            print(f"sub {sub} ({r.scene_info()['n_prims']} prims, traversal {r.set_traversal(-1)}) {side}^2 sampling {mode}: {min(ts)*1e3:7.2f} ms = {side*side*64/min(ts)/1e6:7.1f} Msamples/s ({st.bounce_launches} launches)", flush=True)
