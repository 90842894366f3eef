#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each its own run, no trace domains mixed in).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
cat /sys/fs/cgroup/cpu.max > $OUT/cpu_max.txt 2>&1 || true
nproc > $OUT/nproc.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_trace.log 2>&1
echo trace-done
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_IFETCH SQ_WAVE_CYCLES"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/bench_pmc_$NAME.log 2>&1 || echo "pmc pass $NAME failed"
  echo pmc-$NAME-done
done
find $OUT -name "*.csv" | head -50
