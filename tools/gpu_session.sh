#!/bin/bash
# One parameterised GPU session script (run ON THE MI355X BOX through `tools/gpu.sh tools/gpu_session.sh <tag> <step> [<step> ...]`);
# it replaces the per-call session scripts of rounds 3-5.  Every step writes under gpurun_out/<tag>/ and prints a one-line summary.
#
#   suite                      python -m pytest tests -x -q -m gpu
#   tests:<-k expression>      the GPU tests selected by a -k expression (use + for spaces: tests:asymmetric+or+two_ranks)
#   bench[:name[:args]]        python bench.py <args> (default: --gpus 1 --steps 20 --warmup 5; args with _ for spaces... or BENCH_<NAME>_ARGS)
#   dist1                      the default command under torch.distributed.run --nproc-per-node 1 (one-rank RCCL communicator)
#   ab:<name>:<tags>[:args]    A/B of prebuilt library variants build_ab/lib_<tag>.so (comma list, each run twice, interleaved) on
#                              `bench.py --no-cpu --no-extras <args, _ for spaces> $AB_ARGS`
#   fuzz:<cases>:<seed0>[:ENV=1,ENV2=x;--extra_args]   tools/fuzz_parity.py (environment assignments, then extra arguments with _ for spaces)
#   profiles[:legs]            tools/r6_profiles.sh (legs comma list: p2p,gicp,vgicp,avgicp,hard,c4)
#   timeline                   kernel timeline of one ICP iteration on the one-rank RCCL path (tools/trace_gaps_dist1.sh)
#   probes                     tools/probes/run_valu_probe.sh + run_gather_probe.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:?tag}; shift
O=gpurun_out/$TAG
mkdir -p $O
summ() { # bench line (the compact driver line of round 6) -> a few lines
  python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); f = r["roofline"]
    line = "%-14s %9.0f reg/s  launch %.4f ms  acc/step %.2f  solve/step %.2f  iters %.3f  hbm %.3f  %s %.3f" % (sys.argv[1], r["value"], f["avg_launch_ms"], f["accumulate_ms_per_step"], f["solve_ms_per_step"], r["config"]["iterations_mean"], f.get("hbm_frac") or 0.0, f.get("limiter"), f.get("limiter_frac") or 0.0)
    if "hard_guess" in r: line += "  hard %.0f" % r["hard_guess"]["value"]
    if "process_wall_s" in r: line += "  wall %.0f s" % r["process_wall_s"]
    print(line, flush=True)
    for k, v in (r.get("configs") or {}).items():
        print("    %-14s %s" % (k, json.dumps(v)[:220]), flush=True)
except Exception as e:  # noqa: BLE001
    print(sys.argv[1], "FAILED", repr(e), flush=True)
PY
}
for step in "$@"; do
  IFS=: read -r kind a b c <<< "$step"
  case $kind in
    suite)
      ( time python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1 ) 2> $O/pytest.time; tail -3 $O/pytest.txt | tr '\n' ' '; grep real $O/pytest.time ;;
    tests)
      python -m pytest tests -x -q -m gpu -k "${a//+/ }" > $O/tests_$(echo "$a" | tr -c 'a-zA-Z0-9\n' '_' | cut -c1-40).txt 2>&1; tail -3 $O/tests_*.txt | tail -3 ;;
    bench)
      name=${a:-default}; v="BENCH_$(echo $name | tr a-z A-Z)_ARGS"; args=${!v:-${b//_/ }}; [ -z "$args" ] && args="--gpus 1 --steps 20 --warmup 5"
      ( time python bench.py $args > $O/$name.json 2> $O/$name.err ) 2> $O/$name.time || tail -5 $O/$name.err
      summ $name $O/$name.json ;;
    dist1)
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --legs none > $O/dist1.json 2> $O/dist1.err || tail -5 $O/dist1.err
      summ dist1 $O/dist1.json ;;
    ab)
      for rep in 1 2; do for L in ${b//,/ }; do
        ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras ${c//_/ } ${AB_ARGS:-} > $O/ab_${a}_${L}_$rep.json 2> $O/ab_${a}_${L}_$rep.err || tail -3 $O/ab_${a}_${L}_$rep.err
        summ "$a/$L" $O/ab_${a}_${L}_$rep.json
      done; done ;;
    sq)   # sq:<name>:<lib tag>[:args]  one SQ counter pass of a prebuilt variant: VALU instructions per wave / per SIMD-cycle, TA busy
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/sq_$a && ELM_LIB=$R/build_ab/lib_$b.so timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/$O/sq_$a -o b -- python $R/bench.py --no-cpu --no-extras --warmup 1 --steps 3 ${c//_/ } > $R/$O/sq_$a.json 2> $R/$O/sq_$a.err
        ELM_LIB=$R/build_ab/lib_$b.so timeout 400 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/$O/ta_$a -o b -- python $R/bench.py --no-cpu --no-extras --warmup 1 --steps 3 ${c//_/ } > $R/$O/ta_$a.json 2> $R/$O/ta_$a.err )
      python - $O/sq_$a/b_results.db $O/ta_$a/b_results.db "$a/$b" <<'PY'
import sqlite3, sys
def load(p):
    per = {}
    c = sqlite3.connect(p)
    for k, did, cn, v in c.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name"):
        per.setdefault(k, {}).setdefault(cn, []).append(v)
    dur = {k: (n, a) for k, n, a in c.execute("select name, count(*), avg(end - start) from kernels group by name")}
    return per, dur
try:
    sq, dur = load(sys.argv[1]); ta, _ = load(sys.argv[2])
    k = max((x for x in sq if "k_accumulate" in x), key=lambda x: len(sq[x]["SQ_WAVES"]))
    m = lambda d, n: sum(d[k][n]) / len(d[k][n])
    cyc = m(sq, "GRBM_GUI_ACTIVE") / 8.0
    cyc2 = m(ta, "GRBM_GUI_ACTIVE") / 8.0
    print("%-12s %s launches %d  %.1f us  VALU/wave %.1f  VALU/SIMD-cycle %.4f  waves/SIMD %.2f  wait_any %.3f  wait_inst %.3f | TA busy %.3f  VMEM_RD/wave %.1f  LDS/wave %.1f  SALU/wave %.1f  L1 acc/CU-cycle %.3f" % (
        sys.argv[3], k.split("(")[0][-34:], len(sq[k]["SQ_WAVES"]), dur[k.split("(")[0] if k.split("(")[0] in dur else k][1] / 1e3 if (k.split("(")[0] in dur or k in dur) else 0.0,
        m(sq, "SQ_INSTS_VALU") / m(sq, "SQ_WAVES"), m(sq, "SQ_INSTS_VALU") / (1024.0 * cyc), 4.0 * m(sq, "SQ_WAVE_CYCLES") / (1024.0 * cyc),
        m(sq, "SQ_WAIT_ANY") / m(sq, "SQ_WAVE_CYCLES"), m(sq, "SQ_WAIT_INST_ANY") / m(sq, "SQ_WAVE_CYCLES"),
        m(ta, "TA_TA_BUSY_sum") / 256.0 / cyc2, m(ta, "SQ_INSTS_VMEM_RD") / m(sq, "SQ_WAVES"), m(ta, "SQ_INSTS_LDS") / m(sq, "SQ_WAVES"), m(ta, "SQ_INSTS_SALU") / m(sq, "SQ_WAVES"),
        m(ta, "TCP_TOTAL_CACHE_ACCESSES_sum") / 256.0 / cyc2), flush=True)
except Exception as e:  # noqa: BLE001
    print(sys.argv[3], "FAILED", repr(e), flush=True)
PY
      rm -rf $O/sq_$a $O/ta_$a ;;
    fuzz)
      fe=${c%%;*}; fa=""; [ "$c" != "${c#*;}" ] && fa=${c#*;}   # c = "ENV=1,ENV2=x;--extra_args" (both parts optional)
      env ${FUZZ_ENV:-} ${fe//,/ } timeout 3000 python tools/fuzz_parity.py --cases $a --seed0 $b ${fa//_/ } > $O/fuzz_$b.txt 2>&1
      echo "fuzz $a cases from seed $b [${fe}] [${fa//_/ }]: $(tail -1 $O/fuzz_$b.txt)"; grep -E "^MISMATCH|^PAIR MISMATCH|^pairs:|singular|raised" $O/fuzz_$b.txt | head -8 ;;
    profiles)
      tools/r6_profiles.sh ${a//,/ } ;;
    timeline)
      tools/trace_gaps_dist1.sh $TAG > $O/timeline.txt 2>&1; tail -25 $O/timeline.txt ;;
    probes)
      tools/probes/run_valu_probe.sh > $O/valu_probe.txt 2>&1; tools/probes/run_gather_probe.sh > $O/gather_probe.txt 2>&1; tail -3 $O/valu_probe.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
