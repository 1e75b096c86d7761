#!/bin/bash
# AddressSanitizer + UBSan over the HOST code (ordering, symbolic phase, records, scheduler, solve) on the CPU: the checker's build of the
# host compiled with -fsanitize=address,undefined into /tmp/asan, driven through the oracle's CPU operators (GPU sanitizers are not
# available on the pool).  Round 4: clean on orderings with / without coordinates at 1 and 6 threads over seven matrix classes and on
# whole factorisations (fem27, elastic3d, kkt).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/pangulu_amd/csrc
mkdir -p /tmp/asan
for f in pg_api pg_analysis pg_ordering pg_preprocess pg_numeric pg_comm pg_sptrsv pg_model pg_scaling pg_comm_rccl pg_comm_ipc pg_check; do
  extra=""; [ $f = pg_api ] && extra="-DPANGULU_AMD_TEST_HOOKS"
  g++ -I/opt/rocm/include -O1 -g -std=c++17 -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -DCALCULATE_TYPE_R64 $extra -c host/$f.cpp -o /tmp/asan/$f.o &
done
wait
g++ -shared -Wl,-Bsymbolic -fsanitize=address,undefined -o /tmp/asan/libpangulu_amd_test_r64.so /tmp/asan/*.o build/r64/pg_hip_platform.o -fopenmp -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -ldl -lpthread
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 python tools/sanitize_host_run.py
