#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for c in c2 c3 c5tile c5tile_packed c5tile_fast c5frame c5frame_packed c5frame_fast; do
  f=1; case $c in c5tile*) f=2;; esac
  bash tools/profile.sh r3h_$c $c $f > gpurun_out/prof_r3h_$c.log 2>&1 || { tail -20 gpurun_out/prof_r3h_$c.log; exit 1; }
  tail -1 gpurun_out/prof_r3h_$c.log | cut -c1-200
done
bash tools/profile_rad.sh r3h > gpurun_out/prof_rad_r3h.log 2>&1 || { tail -20 gpurun_out/prof_rad_r3h.log; exit 1; }
