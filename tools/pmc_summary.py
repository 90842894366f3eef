#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc CSVs of tools/profile.sh into the JSON bench.py reads (profiles/<round>_pmc_<config>.json).

    pmc_summary.py <prof_dir> <config> <frames>

Per kernel: sum and per-dispatch mean of every counter.  For the bounce kernel(s) of the workload, per FRAME (the profile
pass renders exactly <frames> frames with the timing build of the kernel):
  hbm_bytes_per_frame / _per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB - rocprofv3 reports both in KiB;
      MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a coalesced 16-B-per-lane stream ->
      doubled; WRITE_SIZE is exact.  Both derive from the L2's fabric-side request counters, so Infinity-Cache hits are
      included: for a scene that fits the 256 MB Infinity Cache this is an upper bound on HBM bytes ("fabric" bytes).
  The rule is calibrated in the same run on this repo's own streaming kernels with known byte counts
  (ptmi_frame_begin reads 24 B/pixel and writes 88; ptmi_resolve reads 16, writes 15).
Stamped with the SHA-256 of libptmi.so and of the kernel sources (ptmi_buildinfo) so that bench.py can tell a stale profile.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-pathtracer_amd", "python"))
import ptmi_buildinfo  # noqa: E402

root, config, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not k.startswith("ptmi::") and "ptmi_" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ptmi::", "").split("<")[0]
        a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
out = {"config": config, "frames": frames, **ptmi_buildinfo.stamps(),
       "kernels": {k: {c: {"sum": v[0], "dispatches": v[1], "mean": v[0] / v[1]} for c, v in cs.items()} for k, cs in agg.items()}}
cands = [k for k in out["kernels"] if k.startswith("ptmi_bounce")]
if cands:
    dom = max(cands, key=lambda k: out["kernels"][k].get("SQ_WAVE_CYCLES", {"sum": 0})["sum"])
    out["dominant_kernel"] = dom
    # a frame may run more than one build of the bounce kernel (the 8-wave build of the packed walk while there are more
    # waves than wave slots, the 7-wave build for the rest of the frame): the frame's figures are sums over all of them
    out["bounce_kernels"] = sorted(cands)
    b = {}
    for k in cands:
        for c, v in out["kernels"][k].items():
            t = b.setdefault(c, {"sum": 0.0, "dispatches": 0})
            t["sum"] += v["sum"]; t["dispatches"] += v["dispatches"]
    if "FETCH_SIZE" in b and "WRITE_SIZE" in b:
        n = b["FETCH_SIZE"]["dispatches"]
        rd = 2.0 * b["FETCH_SIZE"]["sum"] * 1024.0; wr = b["WRITE_SIZE"]["sum"] * 1024.0
        out["launches_per_frame"] = n / frames
        out["fabric_read_bytes_per_frame"] = rd / frames
        out["hbm_write_bytes_per_frame"] = wr / frames
        out["hbm_bytes_per_frame"] = (rd + wr) / frames
        out["hbm_bytes_per_launch"] = (rd + wr) / n
    if "SQ_INSTS_VALU" in b:
        out["valu_wave_insts_per_frame"] = b["SQ_INSTS_VALU"]["sum"] / frames
    if "TCC_HIT_sum" in b and "TCC_MISS_sum" in b:
        out["tcc_hit_rate"] = round(b["TCC_HIT_sum"]["sum"] / max(b["TCC_HIT_sum"]["sum"] + b["TCC_MISS_sum"]["sum"], 1.0), 4)
    if "SQ_THREAD_CYCLES_VALU" in b and "SQ_ACTIVE_INST_VALU" in b:
        out["valu_lane_utilisation"] = round(b["SQ_THREAD_CYCLES_VALU"]["sum"] / max(64.0 * b["SQ_ACTIVE_INST_VALU"]["sum"], 1.0), 4)
    if "SQ_WAIT_ANY" in b and "SQ_WAVE_CYCLES" in b:
        out["wait_frac"] = round(b["SQ_WAIT_ANY"]["sum"] / max(b["SQ_WAVE_CYCLES"]["sum"], 1.0), 4)
cal = {}
for k, rd, wr in (("ptmi_frame_begin", 24, 88), ("ptmi_resolve", 16, 15)):
    kk = out["kernels"].get(k, {})
    if "FETCH_SIZE" in kk and "WRITE_SIZE" in kk:
        cal[k] = {"fetch_KiB_per_dispatch": kk["FETCH_SIZE"]["mean"], "write_KiB_per_dispatch": kk["WRITE_SIZE"]["mean"],
                  "model_read_B_per_pixel": rd, "model_write_B_per_pixel": wr}
out["calibration_kernels"] = cal
print(json.dumps(out, indent=1))
