#!/bin/bash
# GPU box: kernel-trace stats of the Cornell probe (per-kernel durations incl. the sphere fix kernels)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_probe
mkdir -p $OUT
cd $R
SPP=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o probe --output-format csv -- python3 scripts/cornell_probe.py > $OUT/run.log 2>&1 || exit 1
ls $OUT >> $OUT/run.log
