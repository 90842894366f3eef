#!/bin/bash
# TCC hits / misses of one bench workload under two builds:  tools/pmc_ab.sh <config> <lib name in ab_libs/> ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
CFG=$1; shift
for v in "$@"; do
  OUT=gpurun_out/pmc_ab_${CFG}_$v; rm -rf $OUT; mkdir -p $OUT
  export PTMI_LIB=$PWD/ab_libs/libptmi_$v.so
  rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o pmc -- python3 bench.py --config $CFG --profile-pass --steps 1 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "bounce" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
print("$v", {k: f"{v:.4g}" for k, v in acc.items()}, "hit rate %.4f" % (acc["TCC_HIT_sum"] / max(acc["TCC_HIT_sum"] + acc["TCC_MISS_sum"], 1)))
PY
done
