#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
echo "== shipped"; timeout -k 10 300 python tools/radiosity_probe.py 1 2 3 4
echo "== shipped walk 0"; timeout -k 10 300 python tools/radiosity_probe.py 1 2
for v in pf w5 w7 pfw5; do echo "== $v"; PTMI_LIB=$PWD/ab_libs/libptmi_$v.so timeout -k 10 300 python tools/radiosity_probe.py 3 4 --walk 2; done
