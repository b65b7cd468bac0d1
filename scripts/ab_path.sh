#!/bin/bash
# GPU box: Cornell probe (per-kernel ms) with several builds of the library in one call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_path.log
: > $OUT
for lib in "" "$@"; do
  echo "== lib=${lib:-current}" >> $OUT
  PBRT_MI355_LIB=${lib:+$R/$lib} SPP=16 timeout -k 10 200 python -u scripts/cornell_probe.py 2>/dev/null | grep "glass=True leaf=4" >> $OUT || exit 1
done
