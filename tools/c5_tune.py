#!/usr/bin/env python3
"""A/B timing on the c5tile workload (rank 3 of 8 of the 2048^2 1M-triangle frame): traversal mode x top size x block shape x K."""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, ptmi_scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["3:0:0:0", "4:0:0:0"]     # mode:lds_top_records:streams:segments
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
base = ptmi.HostScene.load(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj")).prims()
sc = ptmi_scenes.tessellated_cornell(base, 256, 128)
r = ptmi.Renderer(0)
r.load_scene_arrays(sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
print(r.scene_info(), flush=True)
ref = None
res = {}
for rd in range(rounds):
    for v in variants:
        mode, rec, blk, seg = (int(x) for x in v.split(":"))
        top = r.set_packed_top(rec if rec else 512)
        eff = r.set_traversal(mode)
        r.set_config(spp=spp, max_depth=8, segments_per_launch=seg, collect_stats=False, streams=blk)
        r.update_resolution(2048, 2048, n_ranks=8, rank=3, row_block=8)
        t0 = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t0
        if rd == 0:
            rad = r.read_image(rgb8=False)[1]
            if ref is None: ref = rad
            else: assert (rad.view(np.uint32) == ref.view(np.uint32)).all(), v
        res.setdefault(v, []).append((dt, st.seconds, st.bounce_kernel_ms, st.bounce_launches, eff, top))
n = 2048 * 256 * spp
for v, xs in res.items():
    best = min(x[0] for x in xs)
    print(f"{v:>16}: mode {xs[0][4]} top {xs[0][5]} best {best*1e3:8.2f} ms -> {n/best/1e6:7.1f} Msamples/s  (all: {' '.join(f'{x[0]*1e3:.1f}' for x in xs)}) launches {xs[-1][3]}", flush=True)
