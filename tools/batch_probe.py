#!/usr/bin/env python3
"""Frame batches vs successive frames: Msamples/s on c2 / c3 / c5tile (tools only; bench.py reports the same)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi
import bench
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c3", "c5tile"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = ptmi.Renderer(0)
for name in names:
    cfg = bench.CONFIGS[name]
    bench.load_scene(r, cfg["scene"])
    t = cfg["tiling"] or (1, 0, 8)
    r.update_resolution(cfg["width"], cfg["height"], n_ranks=t[0], rank=t[1], row_block=t[2])
    r.set_config(spp=cfg["spp"], max_depth=cfg["max_depth"], collect_stats=False)
    n = len(r.local_rows()) * cfg["width"] * cfg["spp"]
    r.render_frame()
    t0 = time.perf_counter()
    for _ in range(K): r.render_frame(want_stats=False)
    seq = time.perf_counter() - t0
    r.render_frames(2)
    t0 = time.perf_counter(); r.render_frames(K, want_stats=False)
    for j in range(K): r.select_frame(j)
    bat = time.perf_counter() - t0
    print(f"{name}: {K} successive frames {seq/K*1e3:8.2f} ms/frame {n*K/seq/1e6:8.1f} Msamples/s | batch of {K} {bat/K*1e3:8.2f} ms/frame {n*K/bat/1e6:8.1f} Msamples/s  ({seq/bat:.3f}x)", flush=True)
