#!/bin/bash
# round-4 GPU call 3: parity on the new build (previous-winner bound, gh-only stage 2, wide ordering, WIDE addressing), hard-guess A/B,
# per-method benches without the instrumentation, fuzz, hard "after" + AVGICP / GICP stall profiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/c3.pytest 2>&1; tail -4 gpurun_out/c3.pytest
run() { # tag env.. -- args..
  local tag=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu --no-extras "$@" > gpurun_out/c3_$tag.json 2> gpurun_out/c3_$tag.err || tail -3 gpurun_out/c3_$tag.err
}
run hard_p0 ELM_PREV_WINNER=0 -- --guess hard --steps 6
run hard_p1 ELM_PREV_WINNER=1 -- --guess hard --steps 6
run easy_p0 ELM_PREV_WINNER=0 --
run easy_p1 ELM_PREV_WINNER=1 --
run gicp X=1 -- --method 1
run vgicp X=1 -- --method 2
run avgicp X=1 -- --method 3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c3_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]
        print("%-34s value %8.0f  iters %.3f  ms/step %.2f  launches %d  avg %.4f ms  ps/unit %.2f" % (f, r["value"], r["config"]["iterations_mean"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], 1e9 * ro["avg_launch_ms"] / ro["units_per_launch"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
for s0 in 0 30000 33000; do timeout 900 python tools/fuzz_parity.py --cases 3000 --seed0 $s0 2>&1 | tail -2; done > gpurun_out/c3_fuzz.txt
ELM_GRID=tiled timeout 600 python tools/fuzz_parity.py --cases 800 --seed0 1000 2>&1 | tail -1 >> gpurun_out/c3_fuzz.txt
ELM_GRID_MAX_BLOCK_BYTES=48 timeout 600 python tools/fuzz_parity.py --cases 800 --seed0 70000 2>&1 | tail -1 >> gpurun_out/c3_fuzz.txt
cat gpurun_out/c3_fuzz.txt
BENCH_ARGS="--guess hard" PROF_STALLS=1 PROF_NO_FINAL=1 PROF_STEPS=2 tools/collect_profiles.sh r04_hard_after 2>&1 | tail -1 | cut -c1-600
BENCH_ARGS="--method 3" PROF_STALLS=1 PROF_NO_FINAL=1 tools/collect_profiles.sh r04_avgicp 2>&1 | tail -1 | cut -c1-400
BENCH_ARGS="--method 1" PROF_STALLS=1 PROF_NO_FINAL=1 tools/collect_profiles.sh r04_gicp 2>&1 | tail -1 | cut -c1-400
