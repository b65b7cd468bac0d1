#!/bin/bash
# development aid: sweep the refill threshold of the persistent traversal waves over bench.py
OUT=gpurun_out/refill_sweep.log
: > $OUT
for T in 16 24 32 40 48 56 64; do
  echo "== refill $T" >> $OUT
  PBRT_MI355_REFILL=$T timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('path %.1f Msamples/s %.1f ms' % (d['value']/1e6, d['ms_per_step']), {k:round(v,1) for k,v in d['config']['kernel_ms'].items()}, 'exact/ray %.2f' % d['config']['exact_tests_per_ray'], '| trace %.0f Mrays/s %.2f ms' % (d['closest_hit']['value'], d['closest_hit']['roofline']['kernel_ms']))" >> $OUT || exit 1
done
