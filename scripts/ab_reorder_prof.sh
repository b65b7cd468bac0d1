#!/bin/bash
# GPU box: where the time of the reordering experiment goes - rocprofv3 kernel stats of bench.py --workload c4 with and without
# PBRT_MI355_REORDER (sort kernels vs closest_kernel)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/reorder_prof
mkdir -p $OUT
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/off -o c4 --output-format csv -- python3 bench.py --workload c4 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/off.log 2>&1 || exit 1
PBRT_MI355_REORDER=${BITS:-5} timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/on -o c4 --output-format csv -- python3 bench.py --workload c4 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/on.log 2>&1 || exit 1
python3 - <<'PY' > $OUT/summary.txt
import csv, glob
for tag in ("off", "on"):
    f = glob.glob(f"gpurun_out/reorder_prof/{tag}/**/*kernel_stats.csv", recursive=True)[0]
    print("==", tag)
    for row in list(csv.DictReader(open(f)))[:12]:
        print("%-110s calls %6s total %10.3f ms avg %9.3f ms" % (row["Name"][:110], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e6))
PY
cat $OUT/summary.txt
