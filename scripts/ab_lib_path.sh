#!/bin/bash
# GPU box: A/B of builds of the library (PBRT_MI355_LIB) on the path workloads: C2 (headline), C4 at 8 spp, C5 at 8 spp; per-class times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_lib_path.log
: > $OUT
for L in "$@"; do
  echo "== $L" >> $OUT
  for W in path c4 c5; do
    EXTRA=""; [ $W = path ] && EXTRA="--no-secondary"; [ $W = c4 ] && EXTRA="--spp 8"
    PBRT_MI355_LIB=$R/pbrt-rs_amd/$L timeout -k 10 300 python -u bench.py --workload $W $EXTRA --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
k={n: round(v['ms'], 2) for n, v in f['roofline']['per_kernel'].items()}
print('$W %.1f M samples/s, %.2f ms/frame; %s' % (d['value']/1e6, d['ms_per_step'], json.dumps(k)))" >> $OUT 2>&1 || exit 1
  done
done
cat $OUT
