#!/bin/bash
# developer wrapper: rebuild the in-tree libraries HERE (hipcc cross-compiles), then run a command on the MI355X box
set -e
make -s -C "$(dirname "$0")/../elimaloc_amd/csrc" 2>&1 | grep -E "error|warning" || true
make -s -C "$(dirname "$0")/../oracle" > /dev/null
T=${GPU_TIMEOUT:-1800}
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
