#!/usr/bin/env python3
"""A/B timing of ptmi_bounce variants in ONE process (interleaved rounds): traversal mode x segments_per_launch."""
import itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi
scene = os.path.join(ROOT, "tests", "golden", "scenes", sys.argv[1] if len(sys.argv) > 1 else "cbox.obj")
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 8
modes = [int(m) for m in (sys.argv[5].split(",") if len(sys.argv) > 5 else "0,1".split(","))]
segs = [int(m) for m in (sys.argv[6].split(",") if len(sys.argv) > 6 else "8".split(","))]
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 3
strips = [int(m) for m in (sys.argv[8].split(",") if len(sys.argv) > 8 else ["0"])]     # wave_tiles values
subdiv = int(sys.argv[9]) if len(sys.argv) > 9 else 0
streams = [int(m) for m in (sys.argv[10].split(",") if len(sys.argv) > 10 else ["0"])]
r = ptmi.Renderer(0); r.load_scene(scene, subdiv); r.update_resolution(side, side)
r.set_traversal(-1, 1 << 30)   # lift the sweep size limit so every mode can be forced
print(r.scene_info())
res = {}
for rd in range(rounds):
    for mode, seg, strip, nstr in itertools.product(modes, segs, strips, streams):
        eff = r.set_traversal(mode, 1 << 30); r.set_config(spp=spp, max_depth=depth, segments_per_launch=seg, collect_stats=False, wave_tiles=strip, streams=nstr)
        r.update_resolution(side, side)
        t0 = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t0
        res.setdefault((eff if mode >= 0 else mode, seg, strip, nstr), []).append((dt, st.seconds, st.bounce_kernel_ms, st.bounce_launches))
for k, v in sorted(res.items()):
    best = min(x[0] for x in v); med = sorted(x[0] for x in v)[len(v) // 2]
    print(f"mode {k[0]} seg {k[1]:6d} tiles {k[2]} streams {k[3]}: wall best {best*1e3:8.2f} ms  median {med*1e3:8.2f} ms  -> {side*side*spp/best/1e6:8.1f} Msamples/s  "
          f"kernel {v[-1][2]:8.2f} ms in {v[-1][3]} launches")
