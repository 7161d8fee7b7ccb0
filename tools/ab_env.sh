#!/bin/bash
# Run ON THE MI355X BOX: A/B of a run-time switch of the SAME library on bench.py --no-cpu --no-extras (each setting twice, interleaved)
#   tools/gpu.sh tools/ab_env.sh <tag> <ENVVAR> <value,value,...> [legs: p2p,gicp,hard,vgicp,avgicp,field0..3]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:?tag}; VAR=${2:?env}; VALS=${3:?values}; LEGS=${4:-p2p,gicp,hard}
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do for leg in ${LEGS//,/ }; do for v in ${VALS//,/ }; do
  case $leg in p2p) A="";; gicp) A="--method 1";; vgicp) A="--method 2";; avgicp) A="--method 3";; hard) A="--guess hard";; field*) A="--world field --method ${leg#field} --batch 1024";; esac
  env $VAR=$v python bench.py --no-cpu --no-extras $A > $O/${leg}_${v}_$rep.json 2> $O/${leg}_${v}_$rep.err || tail -3 $O/${leg}_${v}_$rep.err
  python - "$leg/$VAR=$v" $O/${leg}_${v}_$rep.json <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); f = r["roofline"]
    print("%-28s %9.0f reg/s  launch %.4f ms  acc/step %.2f  iters %.3f  flags %d  index %.0f MB  pose %s" % (sys.argv[1], r["value"], f["avg_launch_ms"], f["accumulate_ms_per_step"],
          r["config"]["iterations_mean"], r["config"].get("map_layout_flags", -1), f.get("index_bytes", 0) / 1e6, r.get("pose_err_vs_cpu", {}).get("max_trans_m")), flush=True)
except Exception as e:  # noqa: BLE001
    print(sys.argv[1], "FAILED", repr(e), flush=True)
PY
done; done; done
