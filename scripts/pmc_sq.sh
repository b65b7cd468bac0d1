#!/bin/bash
# one SQ counter pass over bench.py (args after the tag are passed to bench.py); output gpurun_out/prof/pmc_sq_<tag>/
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof
mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$TAG -o s --output-format csv -- python3 bench.py "$@" > $OUT/pmc_sq_$TAG.log 2>&1
