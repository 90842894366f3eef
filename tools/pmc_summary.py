#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs per kernel: sum and per-dispatch mean of every counter."""
import csv
import collections
import glob
import json
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/pmc_*/pmc_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bounce" in k:
            k = "ptmi_bounce<stats>" if "true>" in k.split("(")[0][-7:] else "ptmi_bounce"
        elif k.startswith("ptmi::"):
            k = k.split("(")[0].replace("ptmi::", "")
        else:
            continue
        a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: {"sum": v[0], "dispatches": v[1], "mean": v[0] / v[1]} for c, v in cs.items()} for k, cs in agg.items()}
b = out.get("ptmi_bounce", {})
if "FETCH_SIZE" in b and "WRITE_SIZE" in b:
    # rocprofv3 reports both in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a
    # coalesced 8/16-B-per-lane stream -> doubled; WRITE_SIZE is exact.  Calibration on this repo's own kernels with a known
    # byte count (ptmi_frame_begin reads 24 B/pixel, writes 88; ptmi_resolve reads 16, writes 15) is printed below.
    out["hbm_bytes_per_launch"] = (2.0 * b["FETCH_SIZE"]["mean"] + b["WRITE_SIZE"]["mean"]) * 1024.0
    out["hbm_read_bytes_per_launch"] = 2.0 * b["FETCH_SIZE"]["mean"] * 1024.0
    out["hbm_write_bytes_per_launch"] = b["WRITE_SIZE"]["mean"] * 1024.0
    cal = {}
    for k, rd, wr in (("ptmi_frame_begin", 24, 88), ("ptmi_resolve", 16, 15)):
        if k in out and "FETCH_SIZE" in out[k]:
            n_px = 1024 * 1024
            cal[k] = {"fetch_reported_over_actual": out[k]["FETCH_SIZE"]["mean"] * 1024.0 / (rd * n_px),
                      "write_reported_over_actual": out[k]["WRITE_SIZE"]["mean"] * 1024.0 / (wr * n_px)}
    out["calibration_1024x1024"] = cal
print(json.dumps(out, indent=1))
