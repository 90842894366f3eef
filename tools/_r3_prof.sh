#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for c in c2 c3 c5tile c5tile_packed c5tile_fast c5frame c5frame_packed c5frame_fast; do
  f=1; case $c in c5tile*) f=2;; esac
  bash tools/profile.sh r3g_$c $c $f > gpurun_out/prof_r3g_$c.log 2>&1 || { tail -20 gpurun_out/prof_r3g_$c.log; exit 1; }
  tail -1 gpurun_out/prof_r3g_$c.log | cut -c1-200
done
bash tools/profile_rad.sh r3g > gpurun_out/prof_rad_r3g.log 2>&1 || { tail -20 gpurun_out/prof_rad_r3g.log; exit 1; }
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_final2.json 2> gpurun_out/r3_bench_final2.err || { tail -20 gpurun_out/r3_bench_final2.err; exit 1; }
