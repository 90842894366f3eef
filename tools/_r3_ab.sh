#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for L in "" uni uni5; do
  if [ -n "$L" ]; then export PTMI_LIB=$PWD/ab_libs/libptmi_$L.so; fi
  echo "== lib ${L:-shipped}"
  timeout -k 10 200 python tools/fast_probe.py 64 "0:3:1:80:0,1:3:1:80:0" 3
done
