#!/bin/bash
# GPU box: C2 at max_depth 1 (one closest-hit launch over the 2^26 primary rays) with several builds of the library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_c2_depth1.log
: > $OUT
for lib in "" "$@"; do
  echo "== lib=${lib:-current}" >> $OUT
  PBRT_MI355_LIB=${lib:+$R/$lib} timeout -k 10 200 python -u bench.py --steps 3 --max-depth ${DEPTH:-1} --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2 %.1f ms kernel_ms %s' % (d['ms_per_step'], {k: round(v['ms'],2) for k,v in d['roofline']['per_kernel'].items()}))" >> $OUT 2>&1 || exit 1
done
cat $OUT
