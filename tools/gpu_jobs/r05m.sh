#!/bin/bash
# round 5: stand-alone GETRF kernels (tools/microbench/bench_getrf.hip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05m}_getrf_standalone.log
: > $OUT
for args in "1 256" "64 256" "256 256" "1 128" "256 128"; do
  timeout 120 tools/microbench/bench_getrf.bin $args 2>&1 | tee -a $OUT
done
GP_STAMPS=1 timeout 120 tools/microbench/bench_getrf.bin 1 256 2>&1 | tee -a $OUT
GP_STAMPS=1 timeout 120 tools/microbench/bench_getrf.bin 1 128 2>&1 | tee -a $OUT
