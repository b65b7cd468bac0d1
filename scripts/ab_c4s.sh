#!/bin/bash
# GPU box: C2 probe + C4-shaped path workload with several builds of the library in one call
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_c4s.log
: > $OUT
for lib in "" "$@"; do
  echo "== lib=${lib:-current}" >> $OUT
  PBRT_MI355_LIB=${lib:+$R/$lib} timeout -k 10 200 python -u bench.py --steps 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2 %.1f Msamples/s (%.1f ms) kernel_ms %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v['ms'],2) for k,v in d['roofline']['per_kernel'].items()}))" >> $OUT 2>&1 || exit 1
  PBRT_MI355_LIB=${lib:+$R/$lib} timeout -k 10 300 python -u bench.py --workload c4 --blobs 4096 --res 1920 --spp 16 --steps 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C4-small %.1f Msamples/s  kernel_ms %s' % (d['value']/1e6, {k: round(v['ms'],1) for k,v in d['roofline']['per_kernel'].items()}))" >> $OUT 2>&1 || exit 1
done
cat $OUT
