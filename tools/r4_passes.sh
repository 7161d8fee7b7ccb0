#!/bin/bash
# counter passes of the shipped P2P / GICP kernels and of the hard set at the operating point (tag = $1)
PASSES_ONLY=1 exec "$(dirname "$0")/r4_final2.sh" "${1:-h}"
