#!/bin/bash
# CPU-side sanitizer runs (never on the GPU box: GPU AddressSanitizer is not available on the pool and gpurun refuses it).
#   1. the oracle (oracle/bnr_oracle.c) under gcc's ASan + UBSan: tests/test_oracle_golden.py
#   2. the library's HOST code (argument validation, bnr_rhat_from_stats / bnr_ess_from_stats, bnr_host_*, callback communicators,
#      the no-GPU error paths) under clang's ASan + UBSan: tests/test_host_cpu.py tests/test_julia_shim.py
# Leak detection is off (the Python interpreter itself leaks at exit); every other report aborts the test run.
set -e
cd "$(dirname "$0")/.."
make -C oracle asan > /dev/null
make -C bayesiannetworkregression.jl_amd/csrc asan > /dev/null
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
echo "== oracle under ASan/UBSan"
BNR_ORACLE_LIB=$PWD/oracle/_san/libbnr_oracle.so LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_oracle_golden.py -x -q -m "not gpu" -p no:cacheprovider
echo "== host code of libbnr_hip under ASan/UBSan"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
BNR_HIP_LIB=$PWD/bayesiannetworkregression.jl_amd/csrc/_san/libbnr_hip.so LD_PRELOAD=$RT \
  python -m pytest tests/test_host_cpu.py tests/test_julia_shim.py -x -q -m "not gpu" -p no:cacheprovider
echo "sanitizer runs clean"
