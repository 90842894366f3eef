#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 900 python -m pytest tests/test_distributed_cpu.py -m gpu -x -v --timeout=300 > gpurun_out/r3_tests_5.log 2>&1 || { tail -60 gpurun_out/r3_tests_5.log; exit 1; }
tail -15 gpurun_out/r3_tests_5.log
echo "== the asynchronous-exchange test with the resolve gate compiled out: must FAIL"
PTMI_LIB=$PWD/ab_libs/libptmi_nogate.so timeout -k 10 300 python -m pytest tests/test_distributed_cpu.py -m gpu -x -q -k asynchronous --timeout=200 > gpurun_out/r3_tests_5_nogate.log 2>&1 && echo "UNEXPECTED PASS" || echo "failed as it must"
grep -E "CORRUPTED|MISMATCH|passed|failed" gpurun_out/r3_tests_5_nogate.log | head -5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 > gpurun_out/r3_tests_5b.log 2>&1 || { tail -40 gpurun_out/r3_tests_5b.log; exit 1; }
tail -3 gpurun_out/r3_tests_5b.log
