#!/bin/bash
# Kernel trace + PMC passes over the radiosity solver (tools/radiosity_probe.py 4: cbox subdivided to 8192 primitives).
# Output under gpurun_out/prof_rad_<tag>/ ; tools/pmc_summary.py-style per-kernel sums in pmc_radiosity.json.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_rad_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/radiosity_probe.py 4 > $OUT/trace_log.txt 2>&1
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 tools/radiosity_probe.py 4 > $OUT/log_$NAME.txt 2>&1 || echo "pmc pass $NAME failed"
  echo pmc-$NAME-done
done
python3 tools/pmc_summary.py $OUT radiosity 2 > $OUT/pmc_radiosity.json
find $OUT/trace -name "*kernel_stats.csv" | head -2
