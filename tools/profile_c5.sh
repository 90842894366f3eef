#!/bin/bash
# PMC passes for the 1M-triangle scene probe (LANE walk, scene in L2/MALL/HBM).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_c5_$1
mkdir -p $OUT
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 tools/c5_probe.py 1024 64 32 > $OUT/log_$NAME.txt 2>&1 || echo "pass $NAME failed"
  echo pmc-$NAME-done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/c5_probe.py 1024 64 32 > $OUT/log_trace.txt 2>&1 || echo "trace failed"
echo trace-done
