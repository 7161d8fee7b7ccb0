#!/bin/bash
# round 6 closing session (run ON THE MI355X BOX, after the counter passes have been merged into profiles/pmc_latest.json and committed):
# gpu suite, the driver's default command (line + full record), the one-rank RCCL line, the device-group line on this GPU ({0, 0}: host
# exchange), the kernel trace of the default workload.   -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/${1:-r06z}
mkdir -p $O
( time python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1 ) 2> $O/pytest.time; tail -3 $O/pytest.txt | tr '\n' ' '; grep real $O/pytest.time
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default_line.json 2> $O/default.err ) 2> $O/default.time; cp bench_full.json $O/default_full.json
wc -c $O/default_line.json; grep real $O/default.time
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --legs none > $O/dist1_line.json 2> $O/dist1.err; cp bench_full.json $O/dist1_full.json
python bench.py --gpus 2 --single-process --devices 0,0 --batch 512 --steps 3 --warmup 1 > $O/group00_line.json 2> $O/group00.err
tools/default_trace.sh > $O/default_trace.txt 2>&1; cp gpurun_out/prof_default/kernel_stats.csv $O/default_kernel_stats.csv; cp gpurun_out/prof_default/bench_trace.json $O/default_bench_under_rocprof.json
python - $O <<'PY'
import json, sys, os
o = sys.argv[1]
for n in ("default_line", "dist1_line", "group00_line"):
    try:
        l = json.load(open(os.path.join(o, n + ".json"))); r = l["roofline"]
        print(f"{n:14s} {l['value']:10.1f} reg/s  launch {r['avg_launch_ms']:.4f} ms  hbm {r.get('hbm_frac')}  {r.get('limiter')} {r.get('limiter_frac')}  x{r.get('traffic_over_compulsory')} compulsory  wall {l.get('process_wall_s')}")
        for k, v in (l.get("configs") or {}).items():
            print("    ", k, json.dumps(v)[:230])
        for k in ("hard_guess", "host_fed", "reference_api", "cpu_baseline", "pose_err_vs_cpu", "single_process", "c_caller_ms"):
            if k in l: print("    ", k, json.dumps(l[k])[:230])
    except Exception as e:  # noqa: BLE001
        print(n, "FAILED", repr(e))
PY
head -4 $O/default_kernel_stats.csv
