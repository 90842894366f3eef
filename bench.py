#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel render loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c5tile|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one renderFrame(): all spp samples of every pixel of the frame (scene, path state and RNG streams resident
in HBM; successive steps are successive frames, RNG state carrying over as in the reference, integrator.h:379).

Workloads (BASELINE.json `configs`):
  c2      configs[1]  cbox.obj 1024x1024, 256 spp, max 8 bounces          - the configuration the metric is quoted on (default)
  c3      configs[2]  cbox_quads.obj 1920x1080, 1024 spp, max 5 bounces
  c5tile  configs[4]  ONE GPU's share (rank 3 of 8, interleaved 8-row blocks) of the 1,048,576-triangle 2048x2048 frame,
                      at 64 of its 2048 spp (the rate does not depend on spp): the HBM-relevant workload of this path
  c5frame configs[4]  the whole 2048x2048 frame of that scene on ONE GPU at 64 spp: what the same kernel does when the GPU
                      is full (an eighth of the frame is 0.52 M pixels = 8192 waves, about one per wave slot)
  c4      configs[3]  cbox.obj 4096x4096, 512 spp, max 5 bounces, rows tiled over the N GPUs (strong scaling)
  c5      configs[4]  the whole 1 M-triangle frame tiled over the N GPUs (strong scaling)
N = 1 default: c2 as the headline line + `extra_configs` (c3, c5tile, c5frame), each with its own roofline block.
N > 1 default: weak scaling of c2 - the same view and spp at side = round(1024*sqrt(N)), so every GPU owns ~1024^2
pixels; rows are dealt to ranks in interleaved row blocks (no data-path collective) and ONE RCCL gather at frame end
(ptmi_gather_frame: ncclSend/ncclRecv behind the C ABI, exact tile sizes, 8-bit image) inside the timed region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the library on the streams it
launches on) and `cpu_baseline` (the oracle, timed on the host cores on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver stack
ROOT = os.path.dirname(os.path.abspath(__file__))


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process becomes the launcher.
    It starts N fresh children - one per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment,
    exactly what torch.distributed.run would give them - relays rank 0's JSON line and exits with the worst child's code.
    Runs before torch or the library is imported: the launcher never touches a GPU, and no process that has is re-executed."""
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args()[0].gpus
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else None, text=True if rank == 0 else None))
    import threading
    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line); sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True); t.start()
    worst = 0
    alive = set(range(n))
    while alive:
        for k in sorted(alive):
            rc = procs[k].poll()
            if rc is None:
                continue
            alive.discard(k)
            if rc != 0:
                worst = max(worst, rc if rc > 0 else 1)
                for j in alive:                          # a rank died: the others would wait at the rendezvous for ever
                    procs[j].terminate()
        time.sleep(0.05)
    t.join(timeout=5)
    sys.exit(worst)


if __name__ == "__main__":
    _self_launch()
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ptmi  # noqa: E402
import ptmi_buildinfo  # noqa: E402
import ptmi_dist  # noqa: E402
import ptmi_scenes  # noqa: E402

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SIMD = 1024                # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9             # max shader clock (MI355X_MICROARCH.md, chip-level parameters)
PROFILE_ROUND = "r04"
# tools/gather_rate.hip on this GPU (profiles/r02_gather_rate.txt): dependent random fetches of 32-byte records, 28 active lanes
# per wave, ~50 VALU instructions between fetches - 2.20e11 /s from a 20 MB table, 1.84e11 /s from a 120 MB one, the same at 3 and
# at 8 waves per SIMD: the ceiling the large-scene walk runs against (its records: 20 MB of nodes, 36 MB of triangles, 16 MB of
# material records)
GATHER_PEAK_RECORDS_PER_S = 2.19e11

CONFIGS = {
    "c2": dict(scene="cbox.obj", width=1024, height=1024, spp=256, max_depth=8, tiling=None, kernel="ptmi_bounce",
               served_from="lds", what="BASELINE configs[1]"),
    "c2_fast": dict(scene="cbox.obj", width=1024, height=1024, spp=256, max_depth=8, tiling=None, kernel="ptmi_bounce_wide", fast=True,
                    served_from="lds/l1", what="BASELINE configs[1] through the opt-in fast tree (ptmi_config.fast_tree)"),
    "c3": dict(scene="cbox_quads.obj", width=1920, height=1080, spp=1024, max_depth=5, tiling=None, kernel="ptmi_bounce",
               served_from="lds", what="BASELINE configs[2]"),
    "c5tile": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=(8, 3, 8), kernel="ptmi_bounce_wide",
                   served_from="l2/mall/hbm", what="BASELINE configs[4], one GPU's share (rank 3 of 8) at 64 of 2048 spp; default walk of large triangle "
                   "scenes: the certified walk - 8-wide tree + a per-ray proof that the reference's walk returns the same hit: bit-identical frames"),
    "c5frame": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=None, kernel="ptmi_bounce_wide",
                    served_from="l2/mall/hbm", what="BASELINE configs[4], the WHOLE frame on one GPU at 64 of 2048 spp (default = certified walk)"),
    "c5tile_packed": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=(8, 3, 8), kernel="ptmi_bounce_phased", traversal=4,
                          served_from="l2/mall/hbm", what="c5tile through the reference's own tree (packed layout): rounds 1-2's walk, node for node the reference's"),
    "c5frame_packed": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=None, kernel="ptmi_bounce_phased", traversal=4,
                           served_from="l2/mall/hbm", what="c5frame through the reference's own tree (packed layout)"),
    "c5tile_fast": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=(8, 3, 8), kernel="ptmi_bounce_wide", fast=True,
                        served_from="l2/mall/hbm", what="c5tile through the opt-in fast tree WITHOUT the certificate (ptmi_config.fast_tree: tolerance mode)"),
    "c5frame_fast": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=None, kernel="ptmi_bounce_wide", fast=True,
                         served_from="l2/mall/hbm", what="c5frame through the opt-in fast tree without the certificate"),
    "c5tile_full": dict(scene="tess1m", width=2048, height=2048, spp=2048, max_depth=8, tiling=(8, 3, 8), kernel="ptmi_bounce_wide", counter_spp=64,
                        profile_of="c5tile", served_from="l2/mall/hbm",
                        what="BASELINE configs[4] at its stated 2048 spp, one GPU's share (rank 3 of 8), default = certified walk"),
    "c5_full": dict(scene="tess1m", width=2048, height=2048, spp=2048, max_depth=8, tiling=None, kernel="ptmi_bounce_wide", counter_spp=64,
                    profile_of="c5frame", served_from="l2/mall/hbm",
                    what="BASELINE configs[4] at its stated size - 2048x2048, 2048 spp = 8.6 G samples - the WHOLE frame on one GPU, default = certified walk"),
    "c4": dict(scene="cbox.obj", width=4096, height=4096, spp=512, max_depth=5, tiling=None, kernel="ptmi_bounce", counter_spp=64,
               served_from="lds", what="BASELINE configs[3] at its stated size - 8.6 G samples - on one GPU"),
    "c5strong": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=None, kernel="ptmi_bounce_wide",
                     served_from="l2/mall/hbm", what="BASELINE configs[4] at 64 of its 2048 spp (default = certified walk: the reference's frame)"),
    "c5strong_packed": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=None, kernel="ptmi_bounce_phased", traversal=4,
                            served_from="l2/mall/hbm", what="BASELINE configs[4] at 64 of its 2048 spp through the reference's own tree (packed layout)"),
    "c5strong_fast": dict(scene="tess1m", width=2048, height=2048, spp=64, max_depth=8, tiling=None, kernel="ptmi_bounce_wide", fast=True,
                          served_from="l2/mall/hbm", what="BASELINE configs[4] at 64 of its 2048 spp through the opt-in fast tree"),
    "c2_cert": dict(scene="cbox.obj", width=1024, height=1024, spp=256, max_depth=8, tiling=None, kernel="ptmi_bounce_wide", traversal=6,
                    served_from="lds/l1", what="BASELINE configs[1] with the certified walk forced (the automatic choice for 32 triangles is the sweep)"),
    "c5": dict(scene="tess1m", width=2048, height=2048, spp=2048, max_depth=8, tiling=None, kernel="ptmi_bounce_wide",
               served_from="l2/mall/hbm", what="BASELINE configs[4]"),
}


def tess1m():
    base = ptmi.HostScene.load(os.path.join(SCENES, "cbox_quads.obj")).prims()
    sc = ptmi_scenes.tessellated_cornell(base, 256, 128, seed=1)          # 16 quads x 256 x 128 cells x 2 triangles
    return (sc["type"], sc["verts"], sc["normal"], sc["bsdf"], sc["Le"])


def load_scene(r, name):
    if name == "tess1m":
        r.load_scene_arrays(*tess1m())
    else:
        r.load_scene(os.path.join(SCENES, name))


def algorithmic_bytes_per_sample(st, quads):
    """rays/sample * (144 + 32*nodes/ray + 36|48*tests/ray + 36*hit_rate) + 24."""
    rays_per_sample = st.rays / st.samples
    return rays_per_sample * (144.0 + 32.0 * st.node_visits / st.rays + (48.0 if quads else 36.0) * st.prim_tests / st.rays
                              + 36.0 * st.hits / st.rays) + 24.0


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, math.ceil(q / per)))
            break
        except Exception:
            continue
    return cores


def cpu_baseline(cfg, width, height, target_seconds=12.0):
    """The oracle (CPU restatement, kind "port") on all host cores, same scene/camera/depth, bounded sample: full frame at
    reduced spp for the Cornell scenes, a band of rows for the 1 M-triangle scene."""
    from oracle_binding import OracleScene, default_camera
    cores = host_cores()
    if cfg["scene"] == "tess1m":
        o = OracleScene.from_arrays(*tess1m())
        y0, y1 = height // 2, height // 2 + 8
    else:
        o = OracleScene.load(os.path.join(SCENES, cfg["scene"]))
        y0, y1 = 0, height
    depth = cfg["max_depth"]
    _, _, st = o.render(default_camera(), width, height, 2, max_depth=depth, n_threads=cores, y0=y0, y1=y1)      # calibrate
    rate = st.samples / max(st.seconds, 1e-9)
    spp = int(max(2, min(cfg["spp"], round(target_seconds * rate / (width * (y1 - y0))))))
    _, _, st = o.render(default_camera(), width, height, spp, max_depth=depth, n_threads=cores, y0=y0, y1=y1)
    return {"value": round(st.samples / st.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{cfg['scene']} {width}x{height} rows {y0}:{y1}, {spp} of {cfg['spp']} spp, max_depth {depth}, "
                      f"{st.samples / 1e6:.1f} Msamples in {st.seconds:.1f} s (oracle/ptmi_oracle.c, OpenMP over rows)"}


def load_profile(name, scale_from=None, scale=1.0):
    """profiles/<round>_pmc_<config>.json (tools/profile.sh + tools/pmc_summary.py), or (None, why).  scale_from: the profile
    of that configuration - the same kernel on the same scene and tile at another spp - with its per-frame byte and instruction
    counts multiplied by `scale` (= the ratio of the samples per frame); per-launch figures are dropped."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_{scale_from or name}.json")
    if not os.path.exists(path):
        return None, "no committed profile"
    try:
        prof = json.load(open(path))
    except Exception as e:
        return None, f"unreadable profile: {e}"
    ok, how = ptmi_buildinfo.profile_is_current(prof)
    if not ok:
        return None, "profile_stale"
    prof["_matched"] = how
    prof["_path"] = os.path.relpath(path, ROOT)
    if scale_from:
        for k in ("hbm_bytes_per_frame", "fabric_read_bytes_per_frame", "hbm_write_bytes_per_frame", "valu_wave_insts_per_frame"):
            if prof.get(k) is not None:
                prof[k] = prof[k] * scale
        prof["hbm_bytes_per_launch"] = None
        prof["_path"] += f" x {scale:g} (samples per frame)"
    return prof, None


def roofline_block(name, cfg, m, exact_workload):
    """roofline of the dominant kernel of one measured workload `m` (see measure())."""
    kernel_s = m["kernel_ms"] * 1e-3
    launches = max(m["launches"], 1)
    steps = m["steps"]
    step_s = m["elapsed"] / steps
    if exact_workload and cfg.get("profile_of"):
        prof, why = load_profile(name, cfg["profile_of"], cfg["spp"] / CONFIGS[cfg["profile_of"]]["spp"])
    else:
        prof, why = load_profile(name) if exact_workload else (None, "workload differs from the profiled one")
    # what this design sends to memory by construction: every queued pixel reads and writes its 88-byte state once per
    # launch (+ 4-byte queue entries in and out) - counted live by the library (ptmi_stats.path_visits)
    state_bytes_per_step = m["visits"] * (88 + 88 + 4 + 4) / steps
    alg_bytes_per_step = m["local_samples_per_step"] * m["bytes_per_sample"]
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": cfg["kernel"],
           "launches_per_step": round(launches / steps, 1), "avg_launch_ms": round(m["kernel_ms"] / launches, 4),
           "concurrent_streams": m["streams"],
           "summed_launch_time_over_frame_time": round(kernel_s / max(m["frame_dev_s"], 1e-12), 4)}
    if prof:
        out["kernel"] = prof.get("dominant_kernel", cfg["kernel"])      # e.g. the 8-wave build of the packed walk on c5frame
        if len(prof.get("bounce_kernels", [])) > 1:
            out["kernels_summed"] = prof["bounce_kernels"]
        hbm_bytes_per_step = prof["hbm_bytes_per_frame"]
        # 2 x FETCH_SIZE + WRITE_SIZE, per launch (guide: gfx950 rule); a profile scaled from another spp has no per-launch figure
        out["traffic"] = round(prof["hbm_bytes_per_launch"], 1) if prof["hbm_bytes_per_launch"] is not None else round(hbm_bytes_per_step / max(launches / steps, 1), 1)
        out["achieved_source"] = f"rocprofv3 PMC ({prof['_path']}, stamp matched by {prof['_matched']})"
        out["profile_stale"] = False
    else:
        hbm_bytes_per_step = state_bytes_per_step
        out["traffic"] = None
        out["achieved_source"] = f"path-state byte model, live launch counts ({why})"
        out["profile_stale"] = why == "profile_stale"
    achieved = hbm_bytes_per_step / step_s / 1e9
    out["achieved"] = round(achieved, 3 if achieved < 1.0 else 1)
    out["frac"] = round(min(achieved / HBM_PEAK_GBS, 1.0), 6 if achieved < 8.0 else 4)      # experiments with few, long launches move little state
    assert achieved > 0.0 and 0.0 <= out["frac"] <= 1.0, out
    # SURVEY 8(d)'s algorithmic figure: NOT compared with the HBM peak when the scene is LDS-resident
    out["algorithmic_bytes_per_sample"] = round(m["bytes_per_sample"], 1)
    out["algorithmic_bytes_per_launch"] = round(alg_bytes_per_step * steps / launches, 1)
    out["algorithmic_GBps"] = round(alg_bytes_per_step / step_s / 1e9, 1)
    out["algorithmic_GBps_per_launch"] = round(alg_bytes_per_step * steps / max(kernel_s, 1e-12) / 1e9, 1)
    out["served_from"] = cfg["served_from"]
    out["path_state_bytes_per_launch"] = round(state_bytes_per_step * steps / launches, 1)
    out["path_state_GBps"] = round(state_bytes_per_step / step_s / 1e9, 1)
    if prof:
        if prof.get("fabric_read_bytes_per_frame") is not None:
            # FETCH_SIZE counts L2 -> fabric requests; Infinity-Cache hits are included, so this bounds HBM reads from above
            out["fabric_read_GBps"] = round(prof["fabric_read_bytes_per_frame"] / step_s / 1e9, 1)
            out["hbm_write_GBps"] = round(prof["hbm_write_bytes_per_frame"] / step_s / 1e9, 1)
        for k in ("tcc_hit_rate", "valu_lane_utilisation", "wait_frac"):
            if prof.get(k) is not None:
                out[k] = prof[k]
        insts = prof.get("valu_wave_insts_per_frame")
        if insts:
            per_simd_per_s = insts / max(m["frame_dev_s"] / steps, 1e-12) / N_SIMD
            clk = CLOCK_HZ / per_simd_per_s
            out["valu"] = {"wave_insts_per_step": insts, "insts_per_simd_per_s": round(per_simd_per_s, 1),
                           "clocks_per_inst_at_2p4GHz": round(clk, 3),
                           # guide: a wave64 VALU instruction issues over 2 cycles -> 1.2e9 per SIMD per second at 2.4 GHz
                           "issue_frac_guide_2clk": round(2.0 / clk, 4),
                           # this repo's microbenchmark (tools/valu_rate*.hip): 2.4 clocks for an all-full-rate stream
                           "issue_frac_measured_floor_2p4clk": round(2.4 / clk, 4)}
    if cfg["served_from"] != "lds":
        # the large-scene walk is a stream of dependent per-lane record fetches (one node, triangle or material record each):
        # its ceiling is the chip's random-fetch rate, measured with tools/gather_rate.hip, not a byte rate
        fetches = m["record_fetches_per_step"] / step_s
        out["gather"] = {"record_fetches_per_s": round(fetches, 1), "peak_measured": GATHER_PEAK_RECORDS_PER_S,
                         "frac": round(fetches / GATHER_PEAK_RECORDS_PER_S, 4),
                         "lds_served_node_visits": m["top_node_visit_share"],
                         "what": "node visits not served from the LDS top + primitive tests + material fetches per second vs tools/gather_rate.hip "
                                 "(profiles/r02_gather_rate.txt: uniform random 32-byte records of a 20 MB table; the walk's upper levels are hotter than that)"}
    # `bound` keeps the contract's vocabulary ("hbm" | "mfma": divergent traversal has no MFMA form); what actually limits the
    # workload, from the same counters: the VALU issue rate (sweep over an LDS-resident scene) or the latency of dependent
    # record fetches (the large-scene walks: most wave cycles are spent waiting, neither HBM bandwidth nor issue slots run out)
    if prof and out.get("valu"):
        issue = out["valu"]["issue_frac_guide_2clk"]
        wait = prof.get("wait_frac") or 0.0
        out["limiter"] = ({"kind": "valu_issue", "frac": issue, "of": "one wave-instruction per 2 clocks per SIMD at 2.4 GHz (MI355X_MICROARCH.md)"}
                          if issue >= wait else
                          {"kind": "dependent_fetch_latency", "wait_frac": wait, "valu_issue_frac": issue,
                           "of": "wave cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES); HBM-side bytes and issue slots are both far from their peaks"})
    out["note"] = ("achieved = HBM bytes per step / step time, frac <= 1 by construction; algorithmic_* follow SURVEY 8(d) and count "
                   "node/triangle/material reads, which are served from " + cfg["served_from"])
    return out


def measure(r, cfg, steps, warmup, run_steps, barrier, segments, reduce_max, pipelined=True):
    """counters frame (untimed, stats build) + warmup + `steps` timed frames of the loaded workload.
    run_steps(k, stats, pipelined) renders k steps (frames) and returns the list of their ptmi_stats."""
    # two counter frames where the timed walk is not the reference's own: (1) the reference's walk (forced: the packed layout) for
    # SURVEY 8(d)'s algorithmic figure, which is defined on the reference's node visits and primitive tests; (2) the walk that is
    # timed (certified / fast tree), for what it really fetches
    fast = bool(cfg.get("fast"))
    # counter_spp: the two counter frames of a full-spp configuration of the 1 M-triangle scene are rendered at that many samples per
    # pixel (per-ray and per-sample figures do not depend on the sample count; the reference's walk at 2048 spp would take 7 s)
    r.set_config(spp=cfg.get("counter_spp", cfg["spp"]), max_depth=cfg["max_depth"], segments_per_launch=segments, collect_stats=True, download_image=False, fast_tree=False)
    auto = r.set_traversal(-1)
    own_walk = fast or (auto == r.CERTIFIED and cfg.get("traversal") is None)
    if auto == r.CERTIFIED:
        r.set_traversal(r.PACKED)
    st_counts = r.render_frame()
    r.set_traversal(cfg.get("traversal", -1))
    st_walk = st_counts
    if own_walk:
        r.set_config(fast_tree=fast)
        st_walk = r.render_frame()
    bytes_per_sample = algorithmic_bytes_per_sample(st_counts, r.scene_info()["n_quads"] > 0)
    r.set_config(collect_stats=False, spp=cfg["spp"])
    counter_scale = cfg["spp"] / cfg.get("counter_spp", cfg["spp"])
    if warmup:
        run_steps(warmup, False, pipelined)
    kernel_ms = 0.0; launches = 0; visits = 0; frame_dev_s = 0.0
    barrier()
    t0 = time.perf_counter()
    for st in run_steps(steps, True, pipelined):
        kernel_ms += st.bounce_kernel_ms; launches += st.bounce_launches; visits += st.path_visits; frame_dev_s += st.seconds
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    n_local_px = len(r.local_rows()) * r.width
    return dict(elapsed=elapsed, steps=steps, kernel_ms=kernel_ms, launches=launches, visits=visits, frame_dev_s=frame_dev_s,
                bytes_per_sample=bytes_per_sample, local_samples_per_step=float(n_local_px) * cfg["spp"],
                streams=r.config.streams or (1 if own_walk and not segments else 2 if n_local_px >= (1 << 18) else 1),      # (refill launches: one chunk)
                record_fetches_per_step=counter_scale * float(st_walk.node_visits - st_walk.top_node_visits + st_walk.prim_tests + st_walk.hits + (st_walk.hits if own_walk and not fast else 0)),
                top_node_visit_share=round(st_walk.top_node_visits / max(st_walk.node_visits, 1), 4),
                counters=dict(rays_per_sample=round(st_counts.rays / st_counts.samples, 3),
                              nodes_per_ray=round(st_counts.node_visits / st_counts.rays, 2),
                              tests_per_ray=round(st_counts.prim_tests / st_counts.rays, 2),
                              **({"walk": "certified" if not fast else "fast tree (no certificate)",
                                  "walk_nodes_per_ray": round(st_walk.node_visits / st_walk.rays, 2), "walk_tests_per_ray": round(st_walk.prim_tests / st_walk.rays, 2),
                                  "cert_chain_per_hit": round(st_walk.cert_chain / max(st_walk.hits, 1), 6),
                                  "cert_fallback_per_hit": round(st_walk.cert_fallback / max(st_walk.hits, 1), 8)} if own_walk else {})))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--no-extra", action="store_true", help="N = 1, config c2: skip the extra_configs (c3, c5tile, c5frame)")
    ap.add_argument("--spp", type=int, default=0, help=argparse.SUPPRESS)          # for quick experiments only
    ap.add_argument("--side", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--segments", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--streams", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--extra-budget", type=float, default=240.0, help="seconds the extra_configs may take in all; the ones left are recorded as skipped")
    ap.add_argument("--gather", choices=("rgb8", "radiance", "both"), default="rgb8", help="what the frame-end gather moves (N > 1)")
    ap.add_argument("--pipeline", action="store_true", help="render the K steps as ONE pipelined batch (ptmi_render_frames) instead of one "
                    "ptmi_render_frame call per step; without it the batch rate is still reported as value_pipelined_batch (N = 1)")
    # profile pass (tools/profile.sh): exactly `steps` frames of the timing build of the kernel, nothing else on the GPU
    ap.add_argument("--profile-pass", action="store_true", help=argparse.SUPPRESS)
    # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks: ranks share devices and the gather
    # goes through host memory with gloo.  Never used for reported numbers.
    ap.add_argument("--rehearse-gloo", action="store_true", help=argparse.SUPPRESS)
    # the same with the frame exchange going through ptmi_gather_frame as on N GPUs (torch's own group on gloo): for a
    # librccl.so.1 that accepts ranks sharing a device - tests/mock_rccl.cpp.  Never used for reported numbers.
    ap.add_argument("--rehearse-shared-gpu", action="store_true", help=argparse.SUPPRESS)
    # exercise the RCCL code path (comm init, gather, barrier) even with one rank; never used for reported numbers
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {n_gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    shared = args.rehearse_gloo or args.rehearse_shared_gpu
    device_index = local_rank % torch.cuda.device_count() if shared else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the control plane only - the 128-byte unique id, and barriers / the max over ranks where the
        # library has no communicator - and runs on gloo: the ONE RCCL communicator of the process is libptmi's own
        # (ptmi_dist_init), which carries the frame gather, the barrier and the reduction of the timed region
        dist.init_process_group("gloo", rank=rank, world_size=world)

    name = args.config
    cfg = dict(CONFIGS[name])
    strong = name in ("c4", "c5", "c5strong", "c5strong_packed", "c5strong_fast")
    if name == "c2" and world > 1:                    # weak scaling of the headline configuration
        cfg["width"] = cfg["height"] = int(round(1024 * math.sqrt(n_gpus)))
    if args.side:
        cfg["width"] = cfg["height"] = args.side
    if args.spp:
        cfg["spp"] = args.spp
    exact = not args.side and not args.spp and not args.segments and not args.streams and world == 1
    row_block = 2 if name == "c2" else 8             # 2-row blocks: every rank gets exactly side/N rows at N = 1, 2, 4, 8

    r = ptmi.Renderer(device_index)
    load_scene(r, cfg["scene"])

    def allocate(c):
        if args.streams:
            r.set_config(streams=args.streams)                  # chunks are dealt at update_resolution
        if c["tiling"] and world == 1:
            n_r, rk, rb = c["tiling"]
            r.update_resolution(c["width"], c["height"], n_ranks=n_r, rank=rk, row_block=rb)      # re-seeds the RNG streams
        else:
            r.update_resolution(c["width"], c["height"], n_ranks=world, rank=rank, row_block=row_block)

    allocate(cfg)
    what = {"rgb8": 1, "radiance": 2, "both": 3}[args.gather]
    rccl = use_dist and not args.rehearse_gloo
    fg = None
    if rccl:
        ptmi_dist.dist_init_from_torch(r, dist)       # ships the ncclUniqueId through the torch process group
    elif use_dist:
        fg = ptmi_dist.FrameGather(dist, cfg["width"], cfg["height"], world, rank, row_block, torch.device("cpu"))

    def exchange():
        # the single exchange of a frame.  ptmi_gather_frame only ENQUEUES the RCCL sends/receives and the row placement
        # (library-owned stream); whatever renders or resolves next waits on the device for the gather that still reads the
        # tile.  Everything outstanding is drained by barrier() before the clock stops.
        if rccl:
            r.gather_frame(0, what)
        elif use_dist:
            rgb, rad = r.read_image()
            n = len(rgb)
            fg.send_rgb[:n] = torch.from_numpy(rgb); fg.send_rad[:n] = torch.from_numpy(rad)
            fg.gather()

    def run_steps(k, stats, pipelined):
        # k steps = k successive frames.  Pipelined (ptmi_render_frames): nothing changes between the frames, so a pixel that
        # has finished frame j goes straight on to frame j + 1 and the stragglers of one frame share the GPU with the head of
        # the next; every frame's image is then produced (ptmi_select_frame = its resolve pass) and exchanged like a
        # separately rendered one.  Frame by frame bit-identical to k ptmi_render_frame calls (tests/test_gpu_parity.py).
        if pipelined and k > 1:
            st = r.render_frames(k, want_stats=stats)
            for j in range(k):
                r.select_frame(j)
                exchange()
            return [st]
        out = []
        for _ in range(k):
            out.append(r.render_frame(want_stats=stats))
            exchange()
        return out

    def barrier():
        if rccl:
            r.gather_wait()
            r.dist_barrier()                          # ptmi_dist_barrier: over the library's communicator, then a device-wide wait
        elif use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(elapsed):
        if rccl:
            return r.dist_allreduce_max(elapsed)      # ptmi_dist_allreduce_max
        if not use_dist:
            return elapsed
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    pipelined = args.pipeline
    if args.profile_pass:
        r.set_config(spp=cfg["spp"], max_depth=cfg["max_depth"], segments_per_launch=args.segments, collect_stats=False, fast_tree=bool(cfg.get("fast")))
        r.set_traversal(cfg.get("traversal", -1))
        run_steps(args.steps, False, pipelined)
        print(json.dumps({"profile_pass": name, "frames": args.steps, **ptmi_buildinfo.stamps()}), flush=True)
        r.close()
        return

    m = measure(r, cfg, args.steps, args.warmup, run_steps, barrier, args.segments, reduce_max, pipelined)
    total_samples = float(cfg["width"]) * cfg["height"] * cfg["spp"] * args.steps if not (cfg["tiling"] and world == 1) \
        else m["local_samples_per_step"] * args.steps
    value = total_samples / m["elapsed"] / 1e6

    # the reference's renderFrame() ends with the D2H of the 8-bit image (application.h:211): same steps again with
    # config.download_image (pinned host image); reported next to `value`, which leaves results on the device
    value_incl_d2h = value_other = None
    if world == 1 and not use_dist:
        r.set_config(download_image=True)
        run_steps(min(args.warmup, 1), False, pipelined)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps, False, pipelined)
        barrier()
        value_incl_d2h = total_samples / (time.perf_counter() - t0) / 1e6
        r.set_config(download_image=False)
        if args.steps > 1:                        # the same steps the other way (pipelined batch <-> one call per step), for comparison
            run_steps(min(args.warmup, 2), False, not pipelined)
            barrier()
            t0 = time.perf_counter()
            run_steps(args.steps, False, not pipelined)
            barrier()
            value_other = total_samples / (time.perf_counter() - t0) / 1e6

    out = None
    if rank == 0:
        tiled = f", rank {cfg['tiling'][1]} of {cfg['tiling'][0]} (interleaved {cfg['tiling'][2]}-row blocks)" if cfg["tiling"] and world == 1 else ""
        out = {
            "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}: {cfg['scene']} {cfg['width']}x{cfg['height']}, {cfg['spp']} spp, max_depth {cfg['max_depth']}{tiled}, "
                                   f"default camera, seed 2023 ({cfg['what']})" +
                                   ("" if world == 1 else f", {world} GPUs, interleaved {row_block}-row blocks + 1 RCCL gather ({args.gather}) per frame"),
                       "name": name, "width": cfg["width"], "height": cfg["height"], "spp": cfg["spp"], "max_depth": cfg["max_depth"],
                       "segments_per_launch": args.segments or "default", "parallelism": f"tile{world}", **m["counters"]},
            "roofline": roofline_block(name, cfg, m, exact),
        }
        out["config"]["frames"] = "pipelined batch (ptmi_render_frames)" if pipelined and args.steps > 1 else "one ptmi_render_frame call per step"
        if value_other is not None:
            out["value_one_call_per_step" if pipelined else "value_pipelined_batch"] = round(value_other, 3)
        if value_incl_d2h is not None:
            out["value_incl_d2h"] = round(value_incl_d2h, 3)
            out["d2h_note"] = "value leaves the frame on the device; value_incl_d2h adds renderFrame()'s 3 B/pixel copy to pinned host memory (application.h:211)"

    # N = 1, headline configuration: the other single-GPU workloads of BASELINE.json, driver-timed in the same run
    if world == 1 and not use_dist and name == "c2" and not args.no_extra and exact:
        extras = []
        loaded = None
        t_extras = time.perf_counter()
        for xname, xsteps in (("c3", 3), ("c4", 1), ("c5tile", 6), ("c5tile_full", 2), ("c5tile_packed", 6), ("c5tile_fast", 6), ("c5frame", 3), ("c5_full", 1),
                              ("c5frame_packed", 2), ("c5frame_fast", 3)):
            t_x = time.perf_counter()
            if t_x - t_extras > args.extra_budget:        # wall-clock guard: the line must come out whatever a box does
                extras.append({"name": xname, "skipped": f"extra_configs had used {t_x - t_extras:.0f} s of their {args.extra_budget:.0f} s"})
                continue
            xcfg = dict(CONFIGS[xname])
            if xcfg["scene"] != loaded:                   # the four 1 M-triangle workloads share one load
                load_scene(r, xcfg["scene"]); loaded = xcfg["scene"]
            allocate(xcfg)
            vs_exact = None
            if (xcfg.get("fast") or (xcfg["scene"] == "tess1m" and not xcfg.get("traversal"))) and xname != "c5_full":      # (c5_full: 7 s through the reference's tree; tests/test_gpu_fullsize.py covers rows of it)
                # what the other tree changes in the result, measured on this very workload: one frame through the reference's own tree
                # (packed layout) and one through the timed walk (certified: must be 0 pixels; fast tree: reported) from the same
                # RNG state (update_resolution re-seeds), compared pixel by pixel
                frames = []
                for fast in (False, True):
                    allocate(xcfg)
                    r.set_traversal(-1 if fast else r.PACKED)                     # against the reference's own tree, node for node
                    r.set_config(spp=xcfg["spp"], max_depth=xcfg["max_depth"], segments_per_launch=0, collect_stats=False, fast_tree=fast and bool(xcfg.get("fast")))
                    r.render_frame(want_stats=False)
                    frames.append(r.read_image(rgb8=False)[1])
                r.set_traversal(-1)
                diff = frames[0].view(np.uint32) != frames[1].view(np.uint32)
                vs_exact = {"pixels": int(frames[0].shape[0] * frames[0].shape[1]), "pixels_differ": int(diff.any(axis=-1).sum()),
                            "max_abs": float(np.abs(frames[0].astype(np.float64) - frames[1]).max()),
                            "rmse": float(np.sqrt(np.mean((frames[0].astype(np.float64) - frames[1]) ** 2))),
                            "what": "float radiance of one frame of this workload, the timed walk vs the reference's own tree (packed layout), same RNG state "
                                    "(certified walk: every hit is proven to be the one the reference's walk returns, or walked by that walk - "
                                    "identical whenever the proof's stated precondition holds, DESIGN.md 4.9 - and measured here; fast tree: bar RMSE < 1e-4)"}
                allocate(xcfg)
            xwarm = 0 if xname in ("c5_full", "c4") else 1     # (their two counter frames have warmed everything a 1 - 3 s frame can warm)
            xm = measure(r, xcfg, xsteps, xwarm, run_steps, barrier, 0, reduce_max, pipelined)
            xsamples = xm["local_samples_per_step"] * xsteps
            tiled = f", rank {xcfg['tiling'][1]} of {xcfg['tiling'][0]}" if xcfg["tiling"] else ""
            extras.append({"name": xname, "workload": f"{xcfg['scene']} {xcfg['width']}x{xcfg['height']}, {xcfg['spp']} spp, max_depth {xcfg['max_depth']}{tiled} ({xcfg['what']})",
                           "value": round(xsamples / xm["elapsed"] / 1e6, 3), "unit": "Msamples/s", "steps": xsteps, "warmup": xwarm,
                           "ms_per_step": round(xm["elapsed"] / xsteps * 1e3, 3), **xm["counters"],
                           "roofline": roofline_block(xname, xcfg, xm, True)})
            if xcfg.get("counter_spp"):
                extras[-1]["counters_from"] = f"counter frames at {xcfg['counter_spp']} spp (per-ray / per-sample figures do not depend on the sample count)"
            extras[-1]["wall_s"] = round(time.perf_counter() - t_x, 1)
            if vs_exact:
                extras[-1]["fast_tree_vs_exact" if xcfg.get("fast") else "certified_vs_reference_tree"] = vs_exact
        r.set_config(fast_tree=False)
        out["extra_configs"] = extras

    # N > 1, default line: `value` stays the weak-scaled headline configuration; BASELINE's own multi-GPU configurations are
    # timed in the same run as STRONG scaling (fixed frame, rows tiled over the N ranks, one RCCL gather per frame): configs[3]
    # (c4, at its full 512 spp) and configs[4] (c5, the 1 M-triangle frame at 64 of its 2048 spp: the default = certified walk,
    # the reference's own tree, and the opt-in fast tree without the certificate; and ONE frame at the full 2048 spp, default walk).  A failure here is recorded in the line and does not cost the headline number.
    if world > 1 and name == "c2" and not args.no_extra and not args.side and not args.spp:
        extras = []
        loaded = cfg["scene"]
        t_extras = time.perf_counter()
        for xname, xsteps in (("c4", 2), ("c5strong", 3), ("c5strong_packed", 3), ("c5strong_fast", 3), ("c5", 1)):
            t_x = time.perf_counter()
            # wall-clock guard, decided by rank 0 for everyone (ranks must stay in step)
            over = reduce_max(1.0 if t_x - t_extras > args.extra_budget else 0.0) > 0.0
            if over:
                if rank == 0:
                    extras.append({"name": xname, "skipped": f"extra_configs had used their {args.extra_budget:.0f} s"})
                continue
            try:
                xcfg = dict(CONFIGS[xname])
                if args.rehearse_gloo or args.rehearse_shared_gpu:     # rehearsal on one shared GPU: same control flow, small frames
                    xcfg.update(width=256, height=256, spp=4); xcfg.pop("counter_spp", None)
                if xcfg["scene"] != loaded:
                    load_scene(r, xcfg["scene"] if not (shared and xcfg["scene"] == "tess1m") else "cbox.obj"); loaded = xcfg["scene"]
                r.update_resolution(xcfg["width"], xcfg["height"], n_ranks=world, rank=rank, row_block=8)
                if fg is not None:
                    fg = ptmi_dist.FrameGather(dist, xcfg["width"], xcfg["height"], world, rank, 8, torch.device("cpu"))
                xm = measure(r, xcfg, xsteps, 1, run_steps, barrier, 0, reduce_max, pipelined)
                xsamples = float(xcfg["width"]) * xcfg["height"] * xcfg["spp"] * xsteps
                if rank == 0:
                    extras.append({"name": xname, "scaling": "strong", "workload": f"{xcfg['scene']} {xcfg['width']}x{xcfg['height']}, {xcfg['spp']} spp, max_depth "
                                   f"{xcfg['max_depth']}, rows tiled over {world} GPUs in interleaved 8-row blocks + 1 RCCL gather ({args.gather}) per frame ({xcfg['what']})",
                                   "value": round(xsamples / xm["elapsed"] / 1e6, 3), "unit": "Msamples/s", "n_gpus": world, "steps": xsteps, "warmup": 1,
                                   "ms_per_step": round(xm["elapsed"] / xsteps * 1e3, 3), "wall_s": round(time.perf_counter() - t_x, 1), **xm["counters"]})
            except Exception as e:                       # noqa: BLE001 - recorded, the run goes on
                if rank == 0:
                    extras.append({"name": xname, "error": f"{type(e).__name__}: {e}"})
                break                                     # ranks must stay in step: nobody goes on after a failure
        r.set_config(fast_tree=False)
        if rank == 0:
            out["extra_configs"] = extras

    if rank == 0:
        if rccl:
            out["rccl_ranks"] = r.dist_comm_count()       # what RCCL itself reports for the communicator the gathers ran on (ncclCommCount)
            assert out["rccl_ranks"] == world, f"RCCL sees {out['rccl_ranks']} ranks, the launcher started {world}"
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(cfg, cfg["width"], cfg["height"])
        print(json.dumps(out), flush=True)
    r.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
