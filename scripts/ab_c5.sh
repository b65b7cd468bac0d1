#!/bin/bash
# GPU box: the C5 workload (area light + homogeneous medium, VolPath) with several builds of the library in one call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_c5.log
: > $OUT
for lib in "" "$@"; do
  echo "== lib=${lib:-current}" >> $OUT
  PBRT_MI355_LIB=${lib:+$R/$lib} timeout -k 10 300 python -u bench.py --workload c5 --steps 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C5 %.1f Msamples/s (%.1f ms) kernel_ms %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v['ms'],2) for k,v in d['roofline']['per_kernel'].items()}))" >> $OUT 2>&1 || exit 1
done
cat $OUT
