#!/usr/bin/env python3
"""Timing probe for the 1M-triangle procedural scene (BASELINE.json configs[4]) on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, ptmi_scenes
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
segs = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["8"])]
cu, cv = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (256, 128)
base = ptmi.HostScene.load(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj")).prims()
sc = ptmi_scenes.tessellated_cornell(base, cu, cv)
r = ptmi.Renderer(0)
t = time.time(); r.load_scene_arrays(sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"]); print("load+upload", time.time() - t, r.scene_info(), "mode", r.set_traversal(-1))
r.update_resolution(side, side)
for seg in segs:
    for rep in range(2):
        r.set_config(spp=spp, max_depth=8, segments_per_launch=seg, collect_stats=(rep == 0))
        t = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t
        print(f"seg {seg} rep {rep}: {dt*1e3:.1f} ms -> {side*side*spp/dt/1e6:.1f} Msamples/s, launches {st.bounce_launches}, kernel {st.bounce_kernel_ms:.1f} ms" +
              (f", rays/sample {st.rays/st.samples:.2f} nodes/ray {st.node_visits/st.rays:.1f} tests/ray {st.prim_tests/st.rays:.1f}" if rep == 0 else ""))
