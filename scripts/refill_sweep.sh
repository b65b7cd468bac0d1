#!/bin/bash
# development aid: sweep the refill threshold of the persistent tree-walk waves (C3 ray sets + the C4-shaped path workload)
OUT=gpurun_out/refill_sweep.log
: > $OUT
for T in 16 24 32 40 48; do
  echo "== refill $T" >> $OUT
  PBRT_MI355_REFILL=$T timeout -k 10 200 python -u bench.py --workload trace --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C3 primary %.0f Mrays/s (%.2f ms) shadow %.0f incoherent %.0f' % (d['value'], d['roofline']['kernel_ms'], d['shadow_any_hit']['value'], d['incoherent_closest_hit']['value']))" >> $OUT 2>&1 || exit 1
  PBRT_MI355_REFILL=$T timeout -k 10 300 python -u bench.py --workload c4 --blobs 4096 --res 1920 --spp 16 --steps 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C4-small %.1f Msamples/s  kernel_ms %s' % (d['value']/1e6, {k: round(v,1) for k,v in d['config']['kernel_ms'].items()}))" >> $OUT 2>&1 || exit 1
done
