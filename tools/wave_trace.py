#!/usr/bin/env python3
"""Where the clocks of a wave of ptmi_bounce_phased go on the 1M-triangle scene, for 1/n of the 2048^2 frame on one GPU.
Needs the experiment build (make -C cuda-pathtracer_amd trace-lib -> ab_libs/libptmi_trace.so, -DPTMI_TRACE_WAVES):
    PTMI_LIB=$PWD/ab_libs/libptmi_trace.so python tools/wave_trace.py [spp=64] [n_ranks,...=128,8,1] [traversal mode=-1]
1/128 of the frame = 512 waves on 1024 SIMDs: every wave alone on its SIMD - the sequential chain without any contention."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi, bench
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = ptmi.Renderer(0)
r.load_scene_arrays(*bench.tess1m())
L = C.CDLL(ptmi.LIB_PATH)
out = (C.c_ulonglong * 16)()
mode = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for n_ranks in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (128, 8, 1)):
    eff = r.set_traversal(mode)
    r.set_config(spp=spp, max_depth=8, collect_stats=False)
    r.update_resolution(2048, 2048, n_ranks=n_ranks, rank=n_ranks // 3, row_block=8)
    r.render_frame(); L.ptmi_trace_read(out)
    t0 = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t0
    assert L.ptmi_trace_read(out) == 0
    walk, shade, nw, ns, lw, ls, tot, waves, mx, pe, alive, nn, node = [int(out[i]) for i in range(13)]
    npx = len(r.local_rows()) * 2048
    print(f"mode {eff} node decisions/wave {nn/waves:.0f} at {node/max(nn,1):.0f} clk, prim decisions/wave {(nw-nn)/waves:.0f} at {(walk-node)/max(nw-nn,1):.0f} clk")
    print(f"1/{n_ranks} ({npx} px) {dt*1e3:.1f} ms {npx*spp/dt/1e6:.1f} Msamples/s launches {st.bounce_launches} | waves {waves} "
          f"cycles/wave {tot/waves:.0f} (max {mx}) walk {walk/tot:.3f} shade {shade/tot:.3f} pro/epilogue {pe/tot:.3f} | "
          f"walk decisions/wave {nw/waves:.0f} at {walk/max(nw,1):.0f} cyc, lanes {lw/max(nw,1):.1f}; shade decisions/wave {ns/waves:.0f} at {shade/max(ns,1):.0f} cyc, lanes {ls/max(ns,1):.1f}; "
          f"living lanes/decision {alive/max(nw+ns,1):.1f}", flush=True)
