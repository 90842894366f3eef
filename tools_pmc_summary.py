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
print(json.dumps(out, indent=1))
