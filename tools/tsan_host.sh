#!/bin/bash
# ThreadSanitizer over the HOST code on the CPU (GPU sanitizers are not available on the pool): the checker's build of the host compiled
# with -fsanitize=thread into /tmp/tsan, driven through the oracle's CPU operators by tests/tsan_worker.py at 1, 2 and 4 ranks over the
# host-staged transport (scheduler thread + launcher thread + sender thread per rank).  Reports go to /tmp/tsan/report.<pid>.
#   tools/tsan_host.sh [spec nb]        (default: fem27_8 32, kkt_8 16)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/pangulu_amd/csrc
mkdir -p /tmp/tsan && rm -f /tmp/tsan/report.*
for f in pg_api pg_analysis pg_ordering pg_preprocess pg_numeric pg_comm pg_sptrsv pg_model pg_scaling pg_comm_rccl pg_comm_ipc pg_check; do
  extra=""; [ $f = pg_api ] && extra="-DPANGULU_AMD_TEST_HOOKS"
  g++ -I/opt/rocm/include -O1 -g -std=c++17 -fPIC -fopenmp -fsanitize=thread -fno-omit-frame-pointer -DCALCULATE_TYPE_R64 $extra -c host/$f.cpp -o /tmp/tsan/$f.o &
done
wait
g++ -shared -Wl,-Bsymbolic -fsanitize=thread -o /tmp/tsan/libpangulu_amd_test_r64.so /tmp/tsan/*.o build/r64/pg_hip_platform.o -fopenmp -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -ldl -lpthread || exit 1
cd $R
run() { # world spec nb
  local W=$1 SPEC=$2 NB=$3 PORT=$((21000 + RANDOM % 4000)) pids=() rc=0
  for r in $(seq 0 $((W - 1))); do
    LD_PRELOAD=$(gcc -print-file-name=libtsan.so) OMP_NUM_THREADS=1 PANGULU_AMD_HOST_THREADS=1 \
      TSAN_OPTIONS="log_path=/tmp/tsan/report halt_on_error=0 second_deadlock_stack=1 history_size=4 ignore_noninstrumented_modules=1" \
      timeout 900 python3 tests/tsan_worker.py $r $W $PORT $SPEC $NB &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p || rc=1; done
  echo "[tsan] world $W $SPEC nb $NB: rc $rc"
}
if [ $# -ge 2 ]; then run 1 $1 $2; run 2 $1 $2; run 4 $1 $2
else
  run 1 fem27_8 32; run 2 fem27_8 32; run 4 fem27_8 32; run 4 kkt_8 16
fi
echo "[tsan] reports: $(ls /tmp/tsan/report.* 2>/dev/null | wc -l) file(s)"
grep -h "SUMMARY: ThreadSanitizer" /tmp/tsan/report.* 2>/dev/null | sort | uniq -c | sort -rn | head -40
