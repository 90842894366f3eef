#!/usr/bin/env python3
"""Timing of the guided sampling modes (grid / MIS) next to BSDF sampling, cbox.obj 1024^2 x 256 spp x depth 8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptmi
from guided_fixtures import synthetic_radiosity_grids
r = ptmi.Renderer(0); r.load_scene(os.path.join(ROOT, "tests/golden/scenes/cbox.obj"))
r.set_radiosity_grids(synthetic_radiosity_grids(32, empty_every=0))
r.update_resolution(1024, 1024)
for mode in (0, 2, 3, 0):
    best = 1e9
    for rep in range(3):
        r.set_config(spp=256, max_depth=8, sampling_mode=mode, collect_stats=(rep == 0))
        t = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t
        if rep == 0: rs = st.rays / st.samples
        best = min(best, dt)
    print(f"sampling_mode {mode}: {best*1e3:.2f} ms -> {1024*1024*256/best/1e6:.0f} Msamples/s, rays/sample {rs:.3f}")
