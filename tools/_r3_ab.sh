#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 600 python -m pytest tests/test_distributed_cpu.py -m gpu -x -q --timeout=600 > gpurun_out/r3_tests_dist.log 2>&1 || { tail -40 gpurun_out/r3_tests_dist.log; exit 1; }
tail -2 gpurun_out/r3_tests_dist.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err || { tail -20 gpurun_out/r3_bench_final.err; exit 1; }
python tools/design_table.py gpurun_out/r3_bench_final.json
for c in c2_cert c2_fast; do timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu --no-extra > gpurun_out/r3_$c.json; python - <<PY
import json; d=json.loads(open("gpurun_out/r3_$c.json").read().strip().splitlines()[-1]); print("$c", d["value"], d["ms_per_step"])
PY
done
