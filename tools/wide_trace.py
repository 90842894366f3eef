#!/usr/bin/env python3
"""Where the time of ptmi_bounce_wide goes on the 1 M-triangle scene: one record per wave (experiment build, -DPTMI_TRACE_WAVES:
make -C cuda-pathtracer_amd trace-lib).  PTMI_LIB=$PWD/ab_libs/libptmi_trace.so python tools/wide_trace.py [spp=64] [share=8] [budget_us:sparse,...]
Prints, per variant: resident waves over time (0.5 ms bins), and per kind of scheduling decision (NODE / PRIM / SHADE) the
decisions per wave, lanes advanced per decision and shader clocks per decision."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import numpy as np, ptmi, bench
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
share = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = (sys.argv[3] if len(sys.argv) > 3 else "0:0").split(",")
binw = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
r = ptmi.Renderer(0)
r.load_scene_arrays(*bench.tess1m())
L = C.CDLL(ptmi.LIB_PATH)
L.ptmi_wide_trace_read.restype = C.c_longlong
cap = 1 << 19
buf = np.zeros((cap, 12), dtype=np.uint32)
def read():
    n = L.ptmi_wide_trace_read(buf.ctypes.data_as(C.POINTER(C.c_uint)), C.c_longlong(cap))
    assert n >= 0
    return buf[:min(n, cap)].copy()
for v in variants:
    budget, sparse = v.split(":")
    os.environ["PTMI_REFILL"] = "1" if "r" in sparse else "0"
    os.environ.pop("PTMI_ORDER", None)
    r.set_config(spp=spp, max_depth=8, collect_stats=False)
    r.update_resolution(2048, 2048, n_ranks=share, rank=min(3, share - 1), row_block=8)
    r.render_frame(); read()
    if "o" in sparse:                                   # 'o': successive frames, so that the traced one runs in the order of the last one's costs
        r.render_frame(); read()
    else:
        os.environ["PTMI_ORDER"] = "0"
        r.update_resolution(2048, 2048, n_ranks=share, rank=min(3, share - 1), row_block=8)
    t0 = time.perf_counter(); st = r.render_frame(); dt = time.perf_counter() - t0
    w = read().astype(np.int64)
    half = w[:, 11] >> 8; w[:, 11] &= 0xff
    start = (w[:, 0] - w[:, 0].min()) & 0xffffffff
    start = start - start.min()
    end = start + w[:, 1]
    T = end.max() * 1e-5      # ms (100 MHz ticks)
    npx = len(r.local_rows()) * 2048
    print(f"== budget {budget} us sparse {sparse}: 1/{share} spp {spp}: {dt*1e3:.2f} ms = {npx*spp/dt/1e6:.1f} Msamples/s, {st.bounce_launches} launches, {len(w)} waves, trace span {T:.2f} ms")
    nb = int(T / binw) + 1
    occ = np.zeros(nb); lanes = np.zeros(nb)
    for b in range(nb):
        lo, hi = b * binw * 1e5, (b + 1) * binw * 1e5
        ov = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None)
        occ[b] = ov.sum() / (hi - lo); lanes[b] = (ov * w[:, 11]).sum() / max(ov.sum(), 1)
    print("  resident waves per %.2f ms bin: " % binw + " ".join(f"{int(x)}" for x in occ))
    print("  starting lanes per resident wave: " + " ".join(f"{x:.0f}" for x in lanes))
    hs = np.sort(np.where(half > 0, start + half, end)) * 1e-5
    print("  time (ms) by which 10 / 25 / 50 / 75 / 90 / 99 % of the waves were less than half full: " + " ".join(f"{hs[int(q * (len(hs) - 1))]:.2f}" for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99)))
    live = np.where(half > 0, half, w[:, 1])
    print(f"  wave time before / after a wave is less than half full: {live.sum() / w[:, 1].sum():.3f} / {1 - live.sum() / w[:, 1].sum():.3f}; after the queue ran dry (first wave end) "
          f"{np.clip(end - end.min(), 0, None).sum() / w[:, 1].sum():.3f} of the wave time")
    print(f"  mean resident waves {(w[:, 1].sum() / (T * 1e5)):.0f}; wave duration mean {w[:, 1].mean()*1e-2:.0f} us, max {w[:, 1].max()*1e-2:.0f} us")
    names = ["NODE", "PRIM", "SHADE"]
    tot_clk = sum(w[:, 8 + k].sum() * 16 for k in range(3))
    for k in range(3):
        n, l, c = w[:, 2 + 2 * k].sum(), w[:, 3 + 2 * k].sum(), w[:, 8 + k].sum() * 16
        print(f"  {names[k]:5s}: {n/len(w):8.1f} decisions/wave, {l/max(n,1):5.1f} lanes/decision, {c/max(n,1):7.0f} clk/decision, share of wave clocks {c/max(tot_clk,1):.3f}")
    sys.stdout.flush()
