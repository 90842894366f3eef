#!/bin/bash
# All profiles of a round on one GPU box: tools/profile_all.sh [tag=all] [configs...]  ->  gpurun_out/prof_<tag>_<config>/, gpurun_out/prof_rad_<tag>/
# (configs: any of c2 c3 c4 c5tile c5tile_packed c5tile_fast c5frame c5frame_packed c5frame_fast radiosity; default: all - more than one gpurun call's 20 minutes)
# (copy pmc_<config>.json -> profiles/rNN_pmc_<config>.json and trace/trace_kernel_stats.csv -> profiles/rNN_kernel_stats_<config>.csv)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
TAG=${1:-all}
shift || true
CONFIGS=${*:-c2 c3 c5tile c5tile_packed c5tile_fast c5frame c5frame_packed c5frame_fast radiosity}
for c in $CONFIGS; do
  if [ $c = radiosity ]; then bash tools/profile_rad.sh $TAG > gpurun_out/prof_rad_$TAG.log 2>&1 || { tail -20 gpurun_out/prof_rad_$TAG.log; exit 1; }; echo radiosity-done; continue; fi
  f=1; case $c in c5tile*) f=3;; esac      # (the first frame after update_resolution runs in image order, the others in cost order)
  bash tools/profile.sh ${TAG}_$c $c $f > gpurun_out/prof_${TAG}_$c.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}_$c.log; exit 1; }
  tail -1 gpurun_out/prof_${TAG}_$c.log | cut -c1-200
done
