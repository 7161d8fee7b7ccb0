#!/bin/bash
# final-build campaign (fused kernels + AVGICP fix-up launch): randomised differential fuzz, all index forms, determinism and host-fed soaks
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{
for s0 in 800000 810000 820000 830000; do echo "--seed0 $s0 (5000 cases)"; timeout 900 python tools/fuzz_parity.py --cases 5000 --seed0 $s0 2>&1 | tail -1; done
echo "ELM_AVG_FIXUP=0 --seed0 840000 (2000)"; ELM_AVG_FIXUP=0 timeout 900 python tools/fuzz_parity.py --cases 2000 --seed0 840000 2>&1 | tail -1
echo "ELM_GRID=tiled --seed0 850000 (2000)"; ELM_GRID=tiled timeout 900 python tools/fuzz_parity.py --cases 2000 --seed0 850000 2>&1 | tail -1
echo "ELM_GRID_MAX_BLOCK_BYTES=48 --seed0 860000 (1500, WIDE block addressing)"; ELM_GRID_MAX_BLOCK_BYTES=48 timeout 900 python tools/fuzz_parity.py --cases 1500 --seed0 860000 2>&1 | tail -1
echo "--kernel lists --seed0 870000 (1000)"; timeout 900 python tools/fuzz_parity.py --cases 1000 --seed0 870000 --kernel lists 2>&1 | tail -1
echo "--radar 1.0 --seed0 880000 (1200)"; timeout 900 python tools/fuzz_parity.py --cases 1200 --seed0 880000 --radar 1.0 2>&1 | tail -2
echo "soak_determinism"; timeout 900 python tools/soak_determinism.py 2>&1 | tail -3
echo "soak_hostfed 40"; timeout 900 python tools/soak_hostfed.py 40 2>&1 | tail -3
} > gpurun_out/r4_soak3.txt 2>&1
cat gpurun_out/r4_soak3.txt
