#!/bin/bash
# CPU-side sanitizer run -- in the BUILD container only, never on a GPU box.
#   bash tools/hostcheck.sh
# Builds gnn-tf_amd/lib/hostcheck/libgnx.so (host code of the C ABI under AddressSanitizer + UBSan; device code uninstrumented)
# and the sanitizer build of oracle/propagate_ref.c, then runs the no-GPU tests that enter them -- every exported symbol, the
# argument / error paths, the halo-plan layout code, the oracle's C port against the numpy restatement -- with the sanitizer
# runtime preloaded into python.  Any report fails the run (halt_on_error, abort_on_error).
set -e
cd "$(dirname "$0")/.."
make -C gnn-tf_amd/csrc HOSTCHECK=1 -j4 > /tmp/hostcheck_build.log 2>&1 || { tail -20 /tmp/hostcheck_build.log; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export GNX_HOSTCHECK=1 GNX_LIBRARY=$PWD/gnn-tf_amd/lib/hostcheck/libgnx.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OMP_NUM_THREADS=4
LD_PRELOAD=$RT python3 -m pytest tests/test_abi.py tests/test_oracle_c.py -x -q -p no:cacheprovider "$@"
