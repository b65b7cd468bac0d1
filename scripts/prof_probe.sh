#!/bin/bash
# GPU box: kernel-trace stats of the Cornell probe (per-kernel durations incl. the second-pass kernels).
# usage: prof_probe.sh [other.so ...]   (each library gets its own run; PBRT_MI355_LIB selects it)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lib in "" "$@"; do
  tag=$(basename "${lib:-current}" .so)
  OUT=$R/gpurun_out/prof_probe_$tag
  mkdir -p $OUT
  export PBRT_MI355_LIB=${lib:+$R/$lib}
  SPP=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o probe --output-format csv -- python3 scripts/cornell_probe.py > $OUT/run.log 2>&1 || exit 1
done
