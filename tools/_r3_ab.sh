#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
PTMI_LIB=$PWD/ab_libs/libptmi_npf.so timeout -k 10 900 python -m pytest tests/test_fast_tree.py -m gpu -x -q --timeout=600 > gpurun_out/r3_pb.log 2>&1 || { tail -40 gpurun_out/r3_pb.log; exit 1; }
tail -2 gpurun_out/r3_pb.log
for lib in "" npf "" npf; do
for c in c5tile c5frame; do PTMI_LIB=${lib:+$PWD/ab_libs/libptmi_$lib.so} timeout -k 10 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu --no-extra > gpurun_out/r3_pb_$c.json; python - <<PY
import json; d=json.loads(open("gpurun_out/r3_pb_$c.json").read().strip().splitlines()[-1]); print("lib=$lib $c", d["value"], d["ms_per_step"])
PY
done; done
