#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 600 python -m pytest tests/test_fast_tree.py -m gpu -x -q -s --timeout=400 > gpurun_out/r3_tests_12.log 2>&1 || { tail -50 gpurun_out/r3_tests_12.log; exit 1; }
grep -E "certified|twice|camera 60|the same through|passed|failed" gpurun_out/r3_tests_12.log
