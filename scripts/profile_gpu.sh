#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes for bench.py; output under gpurun_out/prof/.
# Each profiler run is its own process under a time limit; PMC passes never combine with trace domains other than kernel-trace.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
step() { echo "== $*" >> $OUT/progress.log; }
# C2 twice: on ONE lane (PBRT_MI355_LANES=1: every kernel alone on the chip - the durations bench.py's roofline.kernel_ms must
# agree with; bench.py measures its roofline in such steps), then as the default runs it (two lanes: kernels of two passes overlap)
step stats-path
PBRT_MI355_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_path -o path --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/stats_path.log 2>&1 || exit 1
step stats-path-two-lanes
PBRT_MI355_LANES=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_path2 -o path2 --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/stats_path2.log 2>&1 || exit 1
[ "${PROFILE_STATS_ONLY_PATH:-0}" = 1 ] || {
step stats-trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_trace -o trace --output-format csv -- python3 bench.py --workload trace --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats_trace.log 2>&1 || exit 1
step stats-c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_c4 -o c4 --output-format csv -- python3 bench.py --workload c4 --spp 8 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_c4.log 2>&1 || exit 1
step stats-patch
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_patch -o patch --output-format csv -- python3 bench.py --workload patch --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats_patch.log 2>&1 || exit 1
step stats-c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_c5 -o c5 --output-format csv -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_c5.log 2>&1 || exit 1
}
export PBRT_MI355_LANES=1      # the counter passes: one lane (rocprofv3 serialises the dispatches anyway)
for W in ${PROFILE_WORKLOADS:-path trace patch c4 c5}; do
  case $W in
    path) ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-secondary";;
    trace) ARGS="--workload trace --steps 1 --warmup 0 --no-cpu-baseline --no-secondary";;   # primary rays only: one kernel, one ray set
    patch) ARGS="--workload patch --steps 1 --warmup 0 --no-cpu-baseline --no-secondary";;
    c4) ARGS="--workload c4 --spp 8 --steps 1 --warmup 0 --no-cpu-baseline";;
    c5) ARGS="--workload c5 --steps 1 --warmup 0 --no-cpu-baseline";;
  esac
  step pmc-fetch-$W
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch_$W -o f --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_fetch_$W.log 2>&1 || exit 1
  step pmc-write-$W
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_write_$W -o w --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_write_$W.log 2>&1 || exit 1
  step pmc-sq-$W
  timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$W -o s --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_sq_$W.log 2>&1 || exit 1
  step pmc-lanes-$W
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lanes_$W -o l --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_lanes_$W.log 2>&1 || exit 1
done
# lanes of the C4 closest-hit walk with the queue sorted per bounce (compare pmc_lanes_c4)
[ "${PROFILE_REORDER:-1}" = 1 ] || { step done; exit 0; }
step pmc-lanes-c4-reorder
PBRT_MI355_REORDER=5 timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lanes_c4_reorder -o l --output-format csv -- python3 bench.py --workload c4 --spp 8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_lanes_c4_reorder.log 2>&1 || exit 1
step done
ls -R $OUT | head -60 >> $OUT/progress.log
