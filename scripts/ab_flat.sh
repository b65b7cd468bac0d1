#!/bin/bash
# GPU box: flat-sweep probes with several builds of the library in one call. usage: ab_flat.sh lib1.so lib2.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_flat.log
: > $OUT
for lib in "" "$@"; do
  echo "== lib=${lib:-current}" >> $OUT
  PBRT_MI355_LIB=${lib:+$R/$lib} timeout -k 10 200 python -u scripts/flat_probe.py 2>/dev/null | grep "flat_max=64" >> $OUT || exit 1
  PBRT_MI355_LIB=${lib:+$R/$lib} SPP=16 timeout -k 10 200 python -u scripts/cornell_probe.py 2>/dev/null | grep "leaf=4" >> $OUT || exit 1
done
