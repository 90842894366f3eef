#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for L in "" prio75 prio85; do
  if [ -n "$L" ]; then export PTMI_LIB=$PWD/ab_libs/libptmi_$L.so; fi
  echo "== lib ${L:-shipped}"
  timeout -k 10 200 python tools/fast_probe.py 64 "1:3:1:80:0" 3 8,4 | grep -v n_prims
done
