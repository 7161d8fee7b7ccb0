#!/bin/bash
# config-5 latency as a plain-C caller sees it: examples/stream_harness.c at full size (9 M-point map, 131 072-point raw scans)
set -e
gcc -O2 -std=c11 -Iinclude examples/stream_harness.c -Lelimaloc_amd -lelimaloc_hip -lm -Wl,-rpath,$PWD/elimaloc_amd -o /tmp/stream_harness
/tmp/stream_harness
ELM_HARNESS_GRID=3000 ELM_HARNESS_SCAN=131072 ELM_HARNESS_SCANS=60 /tmp/stream_harness
