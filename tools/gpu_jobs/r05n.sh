#!/bin/bash
# round 5: stand-alone GETRF kernels, band chunk 2 against 3
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05n}_getrf_standalone.log
: > $OUT
for b in bench_getrf.bin bench_getrf3.bin; do
  echo "== $b" | tee -a $OUT
  for args in "1 256" "256 256" "1 128"; do
    timeout 120 tools/microbench/$b $args 2>&1 | tee -a $OUT
  done
  GP_STAMPS=1 timeout 120 tools/microbench/$b 1 256 2>&1 | tee -a $OUT
done
