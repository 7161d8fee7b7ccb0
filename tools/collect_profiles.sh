#!/bin/bash
# Run ON THE MI355X BOX (through gpurun) from the repo root: kernel-trace stats + counter passes of bench.py AT THE OPERATING POINT OF
# THE DEFAULT COMMAND (batch 4096 through 256 slots unless PROF_BATCH / BENCH_ARGS say otherwise).
#   tools/collect_profiles.sh <tag>          -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc.json,bench.json}
# Counter passes are separate runs with --kernel-trace only (one hardware counter set per pass); their databases also carry the
# kernel durations, so every pass's own average launch time is recorded beside its counters.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:-} --no-cpu --no-extras --warmup ${PROF_WARMUP:-1} --steps ${PROF_STEPS:-3} --batch ${PROF_BATCH:-4096}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o b -- python $R/bench.py $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass ta TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pass cls1 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
pass cls2 SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
if [ -n "${PROF_STALLS:-}" ]; then  # where the waves wait (AVGICP / GICP / hard guesses)
  pass st1 SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM
  pass st2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
  pass st3 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
fi
[ -n "${PROF_NO_FINAL:-}" ] || python $R/bench.py ${BENCH_ARGS:-} ${FINAL_ARGS:---no-cpu} > $OUT/bench.json 2> $OUT/bench.err
python $R/tools/summarise_profiles.py $OUT
# the rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back): the summaries are what gets committed
rm -rf $OUT/trace $OUT/pmc_*/
