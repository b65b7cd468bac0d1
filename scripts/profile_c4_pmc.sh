#!/bin/bash
# GPU box: PMC passes of the full-size C4 scene (20 971 520 triangles, 3840x2160) at 8 spp = one pass of 2^26 paths
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_c4
mkdir -p $OUT
cd $R
ARGS="--workload c4 --split hlbvh --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch -o f --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || exit 1
echo fetch-done >> $OUT/progress.log
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_write -o w --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1 || exit 1
echo write-done >> $OUT/progress.log
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o s --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || exit 1
echo sq-done >> $OUT/progress.log
