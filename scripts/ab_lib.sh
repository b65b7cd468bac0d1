#!/bin/bash
# GPU box: A/B of two builds of the library (PBRT_MI355_LIB) on the C3 microbench and the C4-shaped path: ab_lib.sh <libA> <libB>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_lib.log
: > $OUT
for L in "$@"; do
  for rep in 1 2; do
  echo "== $L (run $rep)" >> $OUT
  PBRT_MI355_LIB=$R/pbrt-rs_amd/$L timeout -k 10 250 python -u bench.py --workload trace --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
print('C3 primary %.0f Mrays/s (%.3f ms) shadow %.0f incoherent %.0f' % (d['value'], d['roofline']['kernel_ms'], f['shadow_any_hit']['value'], f['incoherent_closest_hit']['value']))" >> $OUT 2>&1 || exit 1
  PBRT_MI355_LIB=$R/pbrt-rs_amd/$L timeout -k 10 250 python -u bench.py --workload c4 --spp 8 --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
k={n: round(v['ms'], 2) for n, v in f['roofline']['per_kernel'].items()}
print('C4 %.1f M samples/s, %.2f ms/frame; %s' % (d['value']/1e6, d['ms_per_step'], json.dumps(k)))" >> $OUT 2>&1 || exit 1
  done
done
cat $OUT
