#!/bin/bash
# round 5, GPU session 1: the antisymmetric side records -- parity tests + the rate on a 10 M-point map that holds such records
set -u
mkdir -p gpurun_out/r5c1
O=gpurun_out/r5c1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "asymmetric or rank_deficient or fix_up or ordinary_maps or fuzz_case or radar_covariance_matches or two_ranks_on_one_gpu" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for m in 1 2 3; do
  python bench.py --method $m --no-cpu --no-extras --steps 10 --warmup 2 > $O/clean_m$m.json 2> $O/clean_m$m.err || tail -3 $O/clean_m$m.err
  python bench.py --method $m --no-cpu --no-extras --steps 10 --warmup 2 --asym-triples 300 > $O/asym_m$m.json 2> $O/asym_m$m.err || tail -3 $O/asym_m$m.err
done
python - <<'PY'
import json
for m in (1,2,3):
    for k in ("clean","asym"):
        try:
            r = json.load(open(f"gpurun_out/r5c1/{k}_m{m}.json"))
            print(k, m, "value %.0f" % r["value"], "flags", r["config"]["map_layout_flags"], "iters %.3f" % r["config"]["iterations_mean"], "launch ms %.4f" % r["roofline"]["avg_launch_ms"])
        except Exception as e: print(k, m, "failed", e)
PY
