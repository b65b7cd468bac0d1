#!/bin/bash
# GPU box: VERDICT r2 item 4(b) - per-bounce ray reordering on BASELINE config 4 (bench.py --workload c4): the queue of every
# bounce after the first sorted by (Morton code of the origin, direction octant) before the closest-hit walk
# (PBRT_MI355_REORDER=<bits per axis>), against the unsorted queue.  The sort's time is counted in the closest class.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_reorder.log
: > $OUT
SPP=${SPP:-8}
run() {
  echo "== $1" >> $OUT
  shift
  env "$@" timeout -k 10 280 python -u bench.py --workload c4 --spp $SPP --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
k={n: round(v['ms'], 2) for n, v in f['roofline']['per_kernel'].items()}
print('%.1f M samples/s, %.2f ms/frame; per class ms: %s' % (d['value']/1e6, d['ms_per_step'], json.dumps(k)))" >> $OUT 2>&1 || exit 1
}
run "unsorted queue (default)" PBRT_MI355_X=0
run "REORDER 5 bits/axis, octant low" PBRT_MI355_REORDER=5
run "REORDER 7 bits/axis, octant low" PBRT_MI355_REORDER=7
run "REORDER 9 bits/axis, octant low" PBRT_MI355_REORDER=9
run "REORDER 7 bits/axis, octant high" PBRT_MI355_REORDER=7 PBRT_MI355_REORDER_OCT_HIGH=1
run "REORDER 3 bits/axis, octant high" PBRT_MI355_REORDER=3 PBRT_MI355_REORDER_OCT_HIGH=1
cat $OUT
