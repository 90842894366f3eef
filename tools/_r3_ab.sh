#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout=600 > gpurun_out/r3_tests_full_6.log 2>&1 || { tail -40 gpurun_out/r3_tests_full_6.log; exit 1; }
tail -3 gpurun_out/r3_tests_full_6.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_final3.json 2> gpurun_out/r3_bench_final3.err || { tail -20 gpurun_out/r3_bench_final3.err; exit 1; }
python tools/design_table.py gpurun_out/r3_bench_final3.json
timeout -k 10 300 python tools/c5_scale.py 64 6 8,4,2,1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
