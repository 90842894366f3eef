import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi, bench
r = ptmi.Renderer(0)
r.load_scene_arrays(*bench.tess1m())
for n_ranks in (8, 1):
  for depth, seg in ((1, 32), (1, 1), (2, 1), (2, 32), (8, 1), (8, 2), (8, 4), (8, 32)):
    r.set_config(spp=64, max_depth=depth, segments_per_launch=seg, collect_stats=True)
    r.update_resolution(2048, 2048, n_ranks=n_ranks, rank=n_ranks // 3, row_block=8)
    st = r.render_frame()
    rays, nv = st.rays, st.node_visits
    r.set_config(collect_stats=False)
    r.render_frame()
    t0 = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t0
    print(f"1/{n_ranks} depth {depth} K {seg}: {dt*1e3:8.2f} ms launches {st.bounce_launches} rays {rays/1e6:.1f} M = {rays/dt/1e9:.3f} Grays/s, nodes/ray {nv/rays:.1f}, node visits/s {nv/dt/1e9:.1f} G", flush=True)
