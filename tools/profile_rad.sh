#!/bin/bash
# PMC pass over the radiosity solver (tools/radiosity_probe.py 4): VALU issue / lane utilisation of ptmi_form_factors.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_radpmc
mkdir -p $OUT
for PASS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 tools/radiosity_probe.py 4 > $OUT/log_$NAME.txt 2>&1 || echo "pmc pass $NAME failed"
  echo pmc-$NAME-done
done
find $OUT -name "*counter_collection.csv"
