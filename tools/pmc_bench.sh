#!/bin/bash
# Run ON THE MI355X BOX: SQ / cache counter passes over a short bench run, printed per accumulate kernel (developer diagnostics).
#   tools/pmc_bench.sh <tag> [bench args]
TAG=${1:-p}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o b -- python $R/bench.py --no-cpu --no-extras --warmup 0 --steps 4 "$@" > $OUT/p$i.log 2> $OUT/p$i.err
done
python - $OUT <<'PY'
import sqlite3, sys, glob, os
out = sys.argv[1]
tot = {}
for db in sorted(glob.glob(os.path.join(out, "p*", "*.db"))):
    c = sqlite3.connect(db)
    for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "k_accumulate" in k:
            tot.setdefault(k.split("(")[0], {})[cn] = (n, avg)
for k, v in tot.items():
    print(k)
    for cn, (n, avg) in sorted(v.items()):
        print(f"   {cn:40s} {avg:18.1f}  ({n} launches)")
    w = v.get("SQ_WAVES", (0, 1))[1]
    if "SQ_INSTS_VALU" in v:
        print(f"   per wave: VALU {v['SQ_INSTS_VALU'][1]/w:.0f}  VMEM_RD {v['SQ_INSTS_VMEM_RD'][1]/w:.1f}  SALU {v['SQ_INSTS_SALU'][1]/w:.0f}  LDS {v['SQ_INSTS_LDS'][1]/w:.1f}")
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"][1] / 8.0  # summed over the 8 XCDs
        print(f"   VALU busy {4*v['SQ_ACTIVE_INST_VALU'][1]/(1024*cyc):.3f} of SIMD cycles (x4 cycles per wave64 instruction), kernel cycles {cyc:.0f}")
    if "TCC_HIT_sum" in v:
        print(f"   L2 hit {v['TCC_HIT_sum'][1]/(v['TCC_HIT_sum'][1]+v['TCC_MISS_sum'][1]):.3f}")
PY
