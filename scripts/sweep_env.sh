#!/bin/bash
# GPU box: tree-walk workloads under several values of one environment variable.  usage: sweep_env.sh NAME v1 v2 ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
NAME=$1; shift
OUT=gpurun_out/sweep_$NAME.log
: > $OUT
for v in "$@"; do
  echo "== $NAME=$v" >> $OUT
  env $NAME=$v timeout -k 10 200 python -u bench.py --workload trace --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C3 primary %.0f shadow %.0f incoherent %.0f Mrays/s' % (d['value'], d['shadow_any_hit']['value'], d['incoherent_closest_hit']['value']))" >> $OUT 2>&1 || exit 1
  env $NAME=$v timeout -k 10 300 python -u bench.py --workload c4 --blobs 4096 --res 1920 --spp 16 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C4-small %.1f Msamples/s  kernel_ms %s' % (d['value']/1e6, {k: round(v['ms'],1) for k,v in d['roofline']['per_kernel'].items()}))" >> $OUT 2>&1 || exit 1
done
cat $OUT
