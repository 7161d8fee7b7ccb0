#!/bin/bash
# re-tune sweep after the instrumentation left the kernels: prebuilt variants of the kernel constants (ELM_LIB), easy + hard for the P2P ones, GICP for its cap
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for L in cur w8 w6 t3 l2 l8 cur; do
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras > gpurun_out/s_${L}_easy.json 2> gpurun_out/s_${L}.err || tail -3 gpurun_out/s_${L}.err
  python - $L easy gpurun_out/s_${L}_easy.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3])); f = r["roofline"]
print("%-8s %-5s %8.0f reg/s  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], f["avg_launch_ms"]), flush=True)
PY
done
for L in cur l2 l8 w6; do
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras --guess hard --steps 6 > gpurun_out/s_${L}_hard.json 2> gpurun_out/s_${L}.err || tail -3 gpurun_out/s_${L}.err
  python - $L hard gpurun_out/s_${L}_hard.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3])); f = r["roofline"]
print("%-8s %-5s %8.0f reg/s  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], f["avg_launch_ms"]), flush=True)
PY
done
for L in cur gw7 gw6 cur; do
  ELM_LIB=$PWD/build_ab/lib_$L.so python bench.py --no-cpu --no-extras --method 1 > gpurun_out/s_${L}_gicp.json 2> gpurun_out/s_${L}.err || tail -3 gpurun_out/s_${L}.err
  python - $L gicp gpurun_out/s_${L}_gicp.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3])); f = r["roofline"]
print("%-8s %-5s %8.0f reg/s  launch %.4f ms" % (sys.argv[1], sys.argv[2], r["value"], f["avg_launch_ms"]), flush=True)
PY
done
