#!/bin/bash
# Run ON THE MI355X BOX: what one RunRegister-equivalent call costs a plain-C caller (examples/register_latency.c): pageable / page-locked /
# resident, 131 072-point scans against a 9 M-point map, P2P and GICP.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
gcc -O2 -std=c11 -Iinclude examples/register_latency.c -Lelimaloc_amd -lelimaloc_hip -lm -Wl,-rpath,$PWD/elimaloc_amd -o /tmp/register_latency || exit 1
mkdir -p gpurun_out
{ /tmp/register_latency; ELM_LAT_METHOD=1 /tmp/register_latency; } 2>&1 | tee gpurun_out/c_latency.txt
