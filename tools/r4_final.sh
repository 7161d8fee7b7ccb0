#!/bin/bash
# round-4 final GPU call: full gpu suite, the default bench line (all legs), per-method lines, the one-rank RCCL path, the default trace,
# counter passes of the SHIPPED kernels at the operating point (all four methods + the hard set), config-5 C harness, smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/f.pytest 2>&1; tail -4 gpurun_out/f.pytest
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1; tail -1 gpurun_out/f_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f_default.json 2> gpurun_out/f_default.err || tail -5 gpurun_out/f_default.err
for m in 1 2 3; do python bench.py --method $m --no-cpu --no-extras > gpurun_out/f_m$m.json 2> gpurun_out/f_m$m.err; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu > gpurun_out/f_dist1.json 2> gpurun_out/f_dist1.err
tools/default_trace.sh > gpurun_out/f_trace.log 2>&1
PROF_NO_FINAL=1 tools/collect_profiles.sh r04f_p2p 2>&1 | tail -1 | cut -c1-200
BENCH_ARGS="--method 1" PROF_NO_FINAL=1 tools/collect_profiles.sh r04f_gicp 2>&1 | tail -1 | cut -c1-200
BENCH_ARGS="--method 2" PROF_NO_FINAL=1 tools/collect_profiles.sh r04f_vgicp 2>&1 | tail -1 | cut -c1-200
BENCH_ARGS="--method 3" PROF_NO_FINAL=1 tools/collect_profiles.sh r04f_avgicp 2>&1 | tail -1 | cut -c1-200
BENCH_ARGS="--guess hard" PROF_NO_FINAL=1 PROF_STEPS=2 tools/collect_profiles.sh r04f_hard 2>&1 | tail -1 | cut -c1-200
tools/c5_c_harness.sh > gpurun_out/f_c5.log 2>&1; tail -2 gpurun_out/f_c5.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/f_*.json")):
    try:
        r = json.load(open(f)); ro = r["roofline"]; ra = r.get("reference_api", {})
        print("%-28s value %8.0f ms/step %.2f launches %d avg %.4f ms | lat1 %s | refapi %s pinned %s | hard %s | hostfed %s | frac %.3f %s | rccl %s" % (
            f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], r["config"].get("latency_ms_batch1"), ra.get("registrations_per_s"),
            ra.get("page_locked_source", {}).get("registrations_per_s"), r.get("hard_guess", {}).get("value"), r.get("host_fed", {}).get("value"), ro["frac"], ro["bound"], r.get("rccl_ranks")))
    except Exception as e:
        print(f, "FAILED", e)
PY
