#!/bin/bash
# GPU box: only the three PMC passes of the C3 primary-ray trace (see profile_gpu.sh)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
ARGS="--workload trace --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch_trace -o f --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_fetch_trace.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_write_trace -o w --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_write_trace.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq_trace -o s --output-format csv -- python3 bench.py $ARGS > $OUT/pmc_sq_trace.log 2>&1 || exit 1
