#!/bin/bash
# development aid: leaf size of the host-built tree (and the device-built one) on the C3 ray sets and the C4-shaped path workload
OUT=gpurun_out/leaf_sweep.log
: > $OUT
for A in "--max-leaf 4" "--max-leaf 2" "--max-leaf 1" "--split hlbvh"; do
  echo "== $A" >> $OUT
  timeout -k 10 200 python -u bench.py --workload trace --no-cpu-baseline $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C3 primary %.0f Mrays/s (%.2f ms, %.1f nodes %.1f prims/ray, build %.2fs) shadow %.0f incoherent %.0f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['nodes_per_ray'], d['roofline']['prims_per_ray'], d['config']['bvh_build_seconds'], d['shadow_any_hit']['value'], d['incoherent_closest_hit']['value']))" >> $OUT 2>&1 || exit 1
  timeout -k 10 300 python -u bench.py --workload c4 --blobs 4096 --res 1920 --spp 16 --steps 2 --no-cpu-baseline --no-secondary $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C4-small %.1f Msamples/s  kernel_ms %s' % (d['value']/1e6, {k: round(v,1) for k,v in d['config']['kernel_ms'].items()}))" >> $OUT 2>&1 || exit 1
done
