#!/bin/bash
# A/B of kernel variants built into ab_libs/libptmi_<name>.so:  tools/ab_libs.sh <script.py> "<args>" name...
set -e
SCRIPT=$1; ARGS=$2; shift; shift
for v in "$@"; do echo "== $v"; PTMI_LIB=$PWD/ab_libs/libptmi_$v.so timeout -k 10 300 python $SCRIPT $ARGS; done
