#!/usr/bin/env python3
"""Rate of the 1M-triangle scene vs the share of the 2048^2 frame one GPU owns (rank 0 of n_ranks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi, bench
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = ptmi.Renderer(0)
r.load_scene_arrays(*bench.tess1m())
modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 4]            # PHASED vs PACKED; 100 + n = PACKED with an n-record LDS top
import numpy as np
for n_ranks in ([int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (16, 8, 4, 2, 1)):
    ref = None
    for mode in modes:
        seg = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # segments per launch (0 = the library's default)
        top = None
        if mode >= 100: top = mode - 100; r.set_packed_top(top); mode = 4          # 100 + n: packed layout with an n-record LDS top
        r.set_traversal(mode)
        r.set_config(spp=spp, max_depth=8, segments_per_launch=seg, collect_stats=False)
        r.update_resolution(2048, 2048, n_ranks=n_ranks, rank=n_ranks // 3, row_block=8)
        n = len(r.local_rows()) * 2048 * spp
        r.render_frame()
        rad = r.read_image(rgb8=False)[1]
        if ref is None: ref = rad
        else: assert (rad.view(np.uint32) == ref.view(np.uint32)).all(), mode
        t0 = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t0
        print(f"mode {mode} top {top} 1/{n_ranks} of the frame ({len(r.local_rows()) * 2048} px): {dt*1e3:8.2f} ms {n/dt/1e6:8.1f} Msamples/s launches {st.bounce_launches} kernel-ms {st.bounce_kernel_ms:.1f}", flush=True)
