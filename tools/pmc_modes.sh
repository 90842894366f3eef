#!/bin/bash
# PMC comparison of traversal modes on cbox (one frame each): tools/pmc_modes.sh <tag> <modes>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_modes_$1
mkdir -p $OUT
for M in ${2//,/ }; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/m$M -o pmc -- python3 tools/ab.py cbox.obj 1024 256 8 $M 32 1 > $OUT/log_$M.txt 2>&1
done
echo done
