#!/bin/bash
# round-4 closing call (tag j) after the fused pair forms: counter passes of the changed kernels at the operating point, merged into
# profiles/pmc_latest.json on the box so the bench lines carry them; then the gpu suite, smoke, the default line with every leg, the
# default trace, per-method lines, the one-rank RCCL line
TAG=${1:-j}; export TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
PROF_NO_FINAL=1 tools/collect_profiles.sh r04${TAG}_p2p 2>&1 | tail -1 | cut -c1-120
BENCH_ARGS="--method 1" PROF_NO_FINAL=1 tools/collect_profiles.sh r04${TAG}_gicp 2>&1 | tail -1 | cut -c1-120
BENCH_ARGS="--method 3" PROF_NO_FINAL=1 tools/collect_profiles.sh r04${TAG}_avgicp 2>&1 | tail -1 | cut -c1-120
python tools/merge_pmc.py gpurun_out/prof_r04${TAG}_p2p gpurun_out/prof_r04${TAG}_gicp gpurun_out/prof_r04${TAG}_avgicp
python -m pytest tests -q -m gpu > gpurun_out/${TAG}.pytest 2>&1; tail -3 gpurun_out/${TAG}.pytest
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_default.json 2> gpurun_out/${TAG}_default.err || tail -5 gpurun_out/${TAG}_default.err
for m in 1 2 3; do python bench.py --method $m --no-cpu --no-extras > gpurun_out/${TAG}_m$m.json 2> gpurun_out/${TAG}_m$m.err; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/${TAG}_dist1.json 2> gpurun_out/${TAG}_dist1.err
tools/default_trace.sh > gpurun_out/${TAG}_trace.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/%s_*.json" % __import__("os").environ.get("TAG", "j"))):
    try:
        r = json.load(open(f)); ro = r["roofline"]; ra = r.get("reference_api", {})
        print("%-26s value %8.0f ms/step %.2f launches %d avg %.4f ms | lat1 %s | refapi %s pinned %s | hard %s | hostfed %s | frac %.3f %s | rccl %s" % (
            f, r["value"], r["ms_per_step"], ro["launches"], ro["avg_launch_ms"], r["config"].get("latency_ms_batch1"), ra.get("registrations_per_s"),
            ra.get("page_locked_source", {}).get("registrations_per_s"), r.get("hard_guess", {}).get("value"), r.get("host_fed", {}).get("value"), ro["frac"], ro["bound"], r.get("rccl_ranks")))
    except Exception as e:
        print(f, "FAILED", e)
PY
