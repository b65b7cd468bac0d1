#!/bin/bash
# development aid: sweep the number of paths per pass of pm_render (C2 and the C4-shaped path workload)
OUT=gpurun_out/pass_sweep.log
: > $OUT
for P in ${PASSES:-1048576 2097152 4194304 8388608 16777216 33554432}; do
  echo "== paths per pass $P" >> $OUT
  PBRT_MI355_PASS_PATHS=$P timeout -k 10 200 python -u bench.py --steps 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2 %.1f Msamples/s (%.1f ms) passes %d kernel_ms %s' % (d['value']/1e6, d['ms_per_step'], d['config']['passes'], {k: round(v,1) for k,v in d['config']['kernel_ms'].items()}))" >> $OUT 2>&1 || exit 1
  PBRT_MI355_PASS_PATHS=$P timeout -k 10 300 python -u bench.py --workload c4 --blobs 4096 --res 1920 --spp 16 --steps 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C4-small %.1f Msamples/s  kernel_ms %s' % (d['value']/1e6, {k: round(v,1) for k,v in d['config']['kernel_ms'].items()}))" >> $OUT 2>&1 || exit 1
done
