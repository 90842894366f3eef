#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 500 python -m pytest tests/test_fast_tree.py -m gpu -x -q --timeout=300 > gpurun_out/r3_tests_9.log 2>&1 || { tail -40 gpurun_out/r3_tests_9.log; exit 1; }
tail -2 gpurun_out/r3_tests_9.log
timeout -k 10 200 python tools/fast_probe.py 64 "1:3:1:80:0" 3 8,1 | grep -v n_prims
