#!/bin/bash
# rocprofv3 kernel trace (+stats) of one bench.py invocation: scripts/trace_run.sh <tag> <bench args...>
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_$TAG -o t --output-format csv -- python3 bench.py "$@" > $OUT/trace_$TAG.log 2>&1
