#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for L in park6 park7 park8; do
  export PTMI_LIB=$PWD/ab_libs/libptmi_$L.so
  echo "== lib $L"
  timeout -k 10 200 python tools/fast_probe.py 64 "1:3:1:9:0" 3 8,1 | grep -v n_prims
done
