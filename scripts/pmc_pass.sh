#!/bin/bash
# one rocprofv3 counter pass over bench.py: scripts/pmc_pass.sh <tag> "<counters>" <bench args...>; output gpurun_out/prof/pmc_<tag>/
set -u
export TMPDIR=/tmp
TAG=$1; CTRS=$2; shift 2
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof
mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc $CTRS -d $OUT/pmc_$TAG -o c --output-format csv -- python3 bench.py "$@" > $OUT/pmc_$TAG.log 2>&1
