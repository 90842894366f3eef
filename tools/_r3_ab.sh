#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 600 python -m pytest tests/test_fast_tree.py tests/test_radiosity_solver.py -m gpu -x -q -s --timeout=400 > gpurun_out/r3_tests_11.log 2>&1 || { tail -40 gpurun_out/r3_tests_11.log; exit 1; }
grep -E "solver through|passed|failed" gpurun_out/r3_tests_11.log
