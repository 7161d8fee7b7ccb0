#!/bin/bash
# the five cases the first pass of tools/r4_soak3.sh reported, and their blocks again, with the singular-system classification
# (build_ab/lib_r04h.so = the library of commit 176049f: git archive 176049f elimaloc_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/elimaloc_amd/csrc)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{
for s0 in 810000 830000; do echo "--seed0 $s0 (5000 cases)"; timeout 900 python tools/fuzz_parity.py --cases 5000 --seed0 $s0 2>&1 | grep -E "singular-system|MISMATCH|cases agree|singular-metric radar"; done
echo "--radar 1.0 --seed0 880000 (1200)"; timeout 900 python tools/fuzz_parity.py --cases 1200 --seed0 880000 --radar 1.0 2>&1 | grep -E "singular-system|MISMATCH|cases agree|singular-metric radar"
echo "r04h library, --seed0 810000 / 830000 (the same blocks with the build before the fused forms)"
for s0 in 810000 830000; do ELM_LIB=$PWD/build_ab/lib_r04h.so timeout 900 python tools/fuzz_parity.py --cases 5000 --seed0 $s0 2>&1 | grep -E "singular-system|MISMATCH|cases agree"; done
} > gpurun_out/r4_soak4.txt 2>&1
cat gpurun_out/r4_soak4.txt
