#!/bin/bash
# GPU box: tree quality against walk time - the device LBVH (default for large scenes) against the host binned-SAH tree on C3 and C4
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_split.log
: > $OUT
for S in hlbvh sah; do
  echo "== --split $S" >> $OUT
  timeout -k 10 250 python -u bench.py --workload trace --steps 5 --no-cpu-baseline --split $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
print('C3 primary %.0f Mrays/s (%.3f ms) shadow %.0f incoherent %.0f; nodes/ray %.1f prims/ray %.2f; build %.3f s' % (d['value'], d['roofline']['kernel_ms'], f['shadow_any_hit']['value'], f['incoherent_closest_hit']['value'], f['roofline']['nodes_per_ray'], f['roofline']['prims_per_ray'], f['config']['bvh_build_seconds']))" >> $OUT 2>&1 || exit 1
  timeout -k 10 400 python -u bench.py --workload c4 --spp 8 --steps 3 --no-cpu-baseline --split $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
k={n: round(v['ms'], 2) for n, v in f['roofline']['per_kernel'].items()}
print('C4 %.1f M samples/s, %.2f ms/frame; %s; build %.3f s' % (d['value']/1e6, d['ms_per_step'], json.dumps(k), f['config']['bvh_build_seconds']))" >> $OUT 2>&1 || exit 1
done
cat $OUT
