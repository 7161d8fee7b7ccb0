#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer pass over the CPU code (the reference has no sanitizer builds):
#   the oracle (test infrastructure) and the host-only files of the product (csrc/elm_ekf.cpp, elm_io.cpp, elm_glue.cpp: no HIP calls in
#   the functions the CPU tests reach).  Builds into build_ab/ (git-ignored), runs the CPU tests against them.  No GPU needed.
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p build_ab
SAN="-O1 -g -std=c++17 -ffp-contract=off -fPIC -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ $SAN -shared -o build_ab/libelm_oracle_san.so oracle/elm_oracle.cpp
g++ $SAN -shared -o build_ab/libelm_host_san.so elimaloc_amd/csrc/elm_ekf.cpp elimaloc_amd/csrc/elm_io.cpp elimaloc_amd/csrc/elm_glue.cpp tools/san/host_stubs.cpp
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  ELM_ORACLE_LIB=$PWD/build_ab/libelm_oracle_san.so ELM_LIB=$PWD/build_ab/libelm_host_san.so python tools/sanitize_run.py "$@" && rc=0 || rc=$?; rm -f build_ab/libelm_oracle_san.so build_ab/libelm_host_san.so; exit $rc
