#!/bin/bash
# Profiles one bench.py workload on the GPU box:  tools/profile.sh <tag> [config=c2] [frames=1]
#   1. kernel trace + stats of the ordinary bench line (rocprofv3 --kernel-trace --stats)
#   2. PMC passes, each its own run with --pmc only (no trace domain mixed in), over `bench.py --profile-pass`:
#      exactly <frames> frames of the timing build of the kernel.
# Output: gpurun_out/prof_<tag>/{trace,pmc_*}; summary JSON: gpurun_out/prof_<tag>/pmc_<config>.json (copy into profiles/).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=$1; CFG=${2:-c2}; FRAMES=${3:-1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cat /sys/fs/cgroup/cpu.max > $OUT/cpu_max.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu --no-extra > $OUT/bench_trace.log 2>&1
echo trace-done
for PASS in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 bench.py --config $CFG --profile-pass --steps $FRAMES > $OUT/bench_pmc_$NAME.log 2>&1 || echo "pmc pass $NAME failed"
  echo pmc-$NAME-done
done
python3 tools/pmc_summary.py $OUT $CFG $FRAMES > $OUT/pmc_$CFG.json
find $OUT/trace -name "*kernel_stats.csv" | head -3
python3 - <<PY
import json; d = json.load(open("$OUT/pmc_$CFG.json"))
print({k: d[k] for k in ("config", "dominant_kernel", "launches_per_frame", "hbm_bytes_per_launch", "hbm_bytes_per_frame", "tcc_hit_rate", "valu_lane_utilisation", "wait_frac") if k in d})
PY
