#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for c in c2 c3 c5tile c5tile_packed c5tile_fast c5frame c5frame_packed c5frame_fast; do
  f=1; case $c in c5tile*) f=2;; esac
  bash tools/profile.sh r3f_$c $c $f > gpurun_out/prof_r3f_$c.log 2>&1 || { tail -20 gpurun_out/prof_r3f_$c.log; exit 1; }
  tail -1 gpurun_out/prof_r3f_$c.log
done
bash tools/profile_rad.sh r3f > gpurun_out/prof_rad_r3f.log 2>&1 || { tail -20 gpurun_out/prof_rad_r3f.log; exit 1; }
tail -2 gpurun_out/prof_rad_r3f.log
