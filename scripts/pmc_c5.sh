set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/prof_c5; mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lanes_c5 -o l --output-format csv -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_lanes_c5.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq_c5 -o s --output-format csv -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq_c5.log 2>&1 || exit 1
