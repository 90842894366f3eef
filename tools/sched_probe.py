#!/usr/bin/env python3
"""Launch scheduling A/B in one process: tools/sched_probe.py <c2|c3|c5> [spp] [shares=8,1] [variants=0:0,1500:1] [rounds=2]
variant = budget_us:sparse[:segments]  (PTMI_BUDGET_US / PTMI_SPARSE, read by renderFrames at every frame; segments > 0 forces
the old fixed-K loop).  Every variant's frame is compared bit for bit with the first one's (same RNG state)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, ptmi_scenes
kind = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shares = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8,1").split(",")]
variants = (sys.argv[4] if len(sys.argv) > 4 else "0:0,1500:1").split(",")
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
r = ptmi.Renderer(0)
if kind == "c5":
    base = ptmi.HostScene.load(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj")).prims()
    sc = ptmi_scenes.tessellated_cornell(base, 256, 128)
    r.load_scene_arrays(sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])
    W, H, depth = 2048, 2048, 8
elif kind == "c3":
    r.load_scene(os.path.join(ROOT, "tests/golden/scenes/cbox_quads.obj"), 0, False); W, H, depth = 1920, 1080, 5
else:
    r.load_scene(os.path.join(ROOT, "tests/golden/scenes/cbox.obj"), 0); W, H, depth = 1024, 1024, 8
print(r.scene_info(), flush=True)
def setup(share):
    if share > 1: r.update_resolution(W, H, n_ranks=share, rank=min(3, share - 1), row_block=8)
    else: r.update_resolution(W, H)
for share in shares:
    ref = None
    for v in variants:
        f = v.split(":"); budget, sparse, seg = f[0], f[1], int(f[2]) if len(f) > 2 else 0
        streams = int(f[3]) if len(f) > 3 else 0
        os.environ["PTMI_REFILL"] = "1" if sparse.endswith("r") else "0"
        r.set_config(spp=spp, max_depth=depth, segments_per_launch=seg, collect_stats=False, streams=streams)
        setup(share)
        r.render_frame()
        rad = r.read_image(rgb8=False)[1]
        if ref is None: ref = rad
        nd = int((rad.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
        ts = []
        for _ in range(rounds):
            setup(share)
            t0 = time.perf_counter(); s2 = r.render_frame(); ts.append(time.perf_counter() - t0)
        n = W * (H // share) * spp
        print(f"{kind} 1/{share} spp {spp} budget {budget:>6} us sparse {sparse} seg {seg} streams {streams}: best {min(ts)*1e3:9.2f} ms = {n/min(ts)/1e6:8.1f} Msamples/s "
              f"(all: {' '.join(f'{t*1e3:.1f}' for t in ts)}), {s2.bounce_launches} launches, kernel ms {s2.bounce_kernel_ms:.1f}; {nd} px differ from the first variant", flush=True)
