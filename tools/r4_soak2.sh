#!/bin/bash
# closing-build campaign (fused pair forms): randomised differential fuzz on the fused and the nine-entry kernels, determinism soak
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{
for s0 in 0 700000; do echo "--seed0 $s0 (3000 cases)"; timeout 900 python tools/fuzz_parity.py --cases 3000 --seed0 $s0 2>&1 | tail -1; done
echo "ELM_PAIR_NINE=1 --seed0 710000 (1500)"; ELM_PAIR_NINE=1 timeout 900 python tools/fuzz_parity.py --cases 1500 --seed0 710000 2>&1 | tail -1
echo "ELM_GRID=tiled --seed0 720000 (1000)"; ELM_GRID=tiled timeout 900 python tools/fuzz_parity.py --cases 1000 --seed0 720000 2>&1 | tail -1
echo "soak_determinism"; timeout 900 python tools/soak_determinism.py 2>&1 | tail -3
} > gpurun_out/r4_soak2.txt 2>&1
cat gpurun_out/r4_soak2.txt
