#!/bin/bash
# run bench.py once per library variant under build/variants (development aid)
OUT=gpurun_out/variants.log
: > $OUT
for v in "" build/variants/*.so; do
  echo "== ${v:-default}" >> $OUT
  PBRT_MI355_LIB=${v:+$PWD/$v} timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('path %.1f Msamples/s %.1f ms' % (d['value']/1e6, d['ms_per_step']), {k:round(v,1) for k,v in d['config']['kernel_ms'].items()}, '| trace %.0f Mrays/s %.2f ms' % (d['closest_hit']['value'], d['closest_hit']['roofline']['kernel_ms']))" >> $OUT || exit 1
done
