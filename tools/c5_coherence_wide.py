import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/cuda-pathtracer_amd/python")
import ptmi, bench
r = ptmi.Renderer(0)
r.load_scene_arrays(*bench.tess1m())
for n_ranks in (8, 1):
  for depth in (1, 2, 3, 8):
    r.set_config(spp=64, max_depth=depth, collect_stats=True)
    r.update_resolution(2048, 2048, n_ranks=n_ranks, rank=n_ranks // 3, row_block=8)
    st = r.render_frame()
    rays, nv, pt = st.rays, st.node_visits, st.prim_tests
    r.set_config(collect_stats=False)
    r.render_frame()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); st = r.render_frame(); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(f"1/{n_ranks} depth {depth}: {dt*1e3:8.2f} ms launches {st.bounce_launches} rays {rays/1e6:.1f} M = {rays/dt/1e9:.3f} Grays/s, nodes/ray {nv/rays:.1f} tris/ray {pt/rays:.1f}, node visits/s {nv/dt/1e9:.1f} G", flush=True)
