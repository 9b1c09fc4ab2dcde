#!/bin/bash
# The CPU test suite on the sanitizer builds (AddressSanitizer + UBSan) of the library's host code and of the oracle's C
# restatement.  CPU container only: GPU sanitizers are not available on the pool.  Leak checking is off (the interpreter's
# own allocations drown everything); any ASan / UBSan report aborts the run with a non-zero exit code.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/spatialcore_amd/csrc asan -j4 > /dev/null
make -C $root/oracle asan > /dev/null
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $root
LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  SPATIALCORE_HIP_LIB=$root/spatialcore_amd/libspatialcore_hip_asan.so SC_ORACLE_LIB=$root/oracle/liboracle_asan.so \
  python -m pytest tests -m "not gpu" -q -p no:cacheprovider "$@"
