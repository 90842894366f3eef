#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout=600 > gpurun_out/r3_tests_full_3.log 2>&1 || { tail -40 gpurun_out/r3_tests_full_3.log; exit 1; }
tail -3 gpurun_out/r3_tests_full_3.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err || { tail -20 gpurun_out/r3_bench_b.err; exit 1; }
python tools/design_table.py gpurun_out/r3_bench_b.json
