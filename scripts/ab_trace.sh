#!/bin/bash
# GPU box: C3 trace bench with two builds of the library in the same call (A/B on one box). usage: ab_trace.sh <other.so>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_trace.log
: > $OUT
for rep in 1 2; do
  for lib in "$1" ""; do
    echo "== lib=${lib:-current} rep=$rep" >> $OUT
    PBRT_MI355_LIB=${lib:+$R/$lib} timeout -k 10 200 python -u bench.py --workload trace --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('primary %.0f Mrays/s (%.2f ms) shadow %.0f incoherent %.0f' % (d['value'], d['roofline']['kernel_ms'], d['shadow_any_hit']['value'], d['incoherent_closest_hit']['value']))" >> $OUT 2>&1 || exit 1
  done
done
