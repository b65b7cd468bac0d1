#!/bin/bash
# GPU box: the bilinear-patch microbench (bench.py --workload patch) with the patch-only walk (PM_WALK_PATCH, round 3) against
# round 2's routing of patch scenes through the MIXED walk (PBRT_MI355_PATCH_WALK=0), device-built and host-built tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=gpurun_out/ab_patch.log
: > $OUT
run() {
  echo "== $1" >> $OUT
  shift
  env "$@" timeout -k 10 250 python -u bench.py --workload patch --steps 5 --no-cpu-baseline ${EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('gpurun_out/bench_full.json'))
print('primary %.0f Mrays/s (%.2f ms) shadow %.0f incoherent %.0f; nodes/ray %.1f prims/ray %.2f; build %s %.3f s' % (d['value'], d['roofline']['kernel_ms'], f['shadow_any_hit']['value'], f['incoherent_closest_hit']['value'], f['roofline']['nodes_per_ray'], f['roofline']['prims_per_ray'], f['config']['bvh_build'], f['config']['bvh_build_seconds']))" >> $OUT 2>&1 || exit 1
}
run "patch-only walk, device LBVH" PBRT_MI355_PATCH_WALK=1
run "MIXED walk (round 2 routing), device LBVH" PBRT_MI355_PATCH_WALK=0
EXTRA="--split sah" run "patch-only walk, host SAH tree" PBRT_MI355_PATCH_WALK=1
EXTRA="--split sah" run "MIXED walk, host SAH tree" PBRT_MI355_PATCH_WALK=0
cat $OUT
