#!/bin/bash
# round 5, GPU session 2: the new bench line (all legs) + the rank-agreement test + the suite's multi-rank tests
set -u
mkdir -p gpurun_out/r5c2
O=gpurun_out/r5c2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default.json 2> $O/default.err ) 2> $O/default.time; tail -3 $O/default.time; tail -5 $O/default.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5c2/default.json"))
print("value %.0f frac %.3f bound %s process_wall %.1f" % (r["value"], r["roofline"]["frac"], r["roofline"]["bound"], r["process_wall_s"]))
for k, v in r.get("configs", {}).items():
    if k == "field_world":
        for m in ("P2P", "GICP", "VGICP", "AVGICP"):
            l = v[m]; print(" field", m, "%.0f vs lattice %.0f (%.2f)" % (l["value"], l["lattice_world_same_shape"], l["vs_lattice_world"]), "flags", l["map_layout_flags"], "tested %.1f" % l["roofline"]["tested_candidates_per_point"], "iters %.2f" % l["iterations_mean"], "succ %.3f" % l["success_rate"], l.get("pose_err_vs_cpu"))
        print(" field wall", v["leg_wall_s"])
    else:
        print(" ", k, "value %.4g %s" % (v["value"], v["unit"]), "wall %.1f" % v["leg_wall_s"], "roof", v["roofline"].get("bound"), v["roofline"].get("frac"), v.get("pose_err_vs_cpu"))
PY
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "disagree or two_ranks or multi_rank or half_set" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
