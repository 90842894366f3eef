#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -e
for T in 0 256; do echo "== PTMI_SOLVER_TOP=$T"; PTMI_SOLVER_TOP=$T timeout -k 10 120 python tools/radiosity_probe.py 4; done
echo "== no grid index (wrong results: upper bound of what the acos/atan2 per unblocked sample cost)"
PTMI_SOLVER_TOP=0 PTMI_LIB=$PWD/ab_libs/libptmi_nogrid.so timeout -k 10 120 python tools/radiosity_probe.py 4
