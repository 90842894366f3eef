#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
timeout -k 10 300 python tools/_r3_exh.py
for c in c2 c2_sweep c3 c3_sweep; do timeout -k 10 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu --no-extra > gpurun_out/r3_exh_$c.json; python - <<PY
import json; d=json.loads(open("gpurun_out/r3_exh_$c.json").read().strip().splitlines()[-1]); print("$c", d["value"], d["ms_per_step"], d["roofline"]["kernel"], {k:v for k,v in d["config"].items() if "walk" in k or "cert" in k})
PY
done
