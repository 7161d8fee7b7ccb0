#!/bin/bash
# Run ON THE MI355X BOX (through gpurun) from the repo root: kernel-trace stats + HBM traffic PMC passes of bench.py.
#   tools/collect_profiles.sh <tag>          -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc.json,bench.json}
# PMC passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:-} --no-cpu --no-extras --warmup 0 --steps 3 --batch ${PROF_BATCH:-1024}"
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o b -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o b -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o b -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
timeout 240 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $OUT/pmc_ta -o b -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_ta.err
python $R/bench.py ${BENCH_ARGS:-} ${FINAL_ARGS:---no-cpu} > $OUT/bench.json 2> $OUT/bench.err
python $R/tools/summarise_profiles.py $OUT
# the rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back): the summaries are what gets committed
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_ta
