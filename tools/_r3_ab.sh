#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout=600 > gpurun_out/r3_tests_full_4.log 2>&1 || { tail -40 gpurun_out/r3_tests_full_4.log; exit 1; }
tail -3 gpurun_out/r3_tests_full_4.log
timeout -k 10 300 python tools/radiosity_probe.py 3 4
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench_c.json 2> gpurun_out/r3_bench_c.err || { tail -20 gpurun_out/r3_bench_c.err; exit 1; }
python tools/design_table.py gpurun_out/r3_bench_c.json
