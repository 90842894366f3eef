#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 900 python -m pytest tests/test_fast_tree.py tests/test_gpu_fullsize.py tests/test_radiosity_solver.py -m gpu -x -q --timeout=600 > gpurun_out/r3_pb.log 2>&1 || { tail -40 gpurun_out/r3_pb.log; exit 1; }
tail -2 gpurun_out/r3_pb.log
timeout -k 10 300 python tools/radiosity_probe.py 2 3 4
timeout -k 10 300 python tools/radiosity_probe.py 4 --fast
