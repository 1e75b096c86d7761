#!/bin/bash
# round 5: dense solves, ring kernel (factor tiles requested ahead through LDS) against the direct kernel (tools/microbench/bench_trsm.hip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05q}_trsm_ring.log
: > $OUT
for args in "1 1 1" "1 0 1" "4 1 1" "64 1 4" "64 0 4" "1024 1 64" "1024 0 64"; do
  for mode in 1 2; do
    echo "== kernel $mode (1 direct, 2 ring) args $args" | tee -a $OUT
    timeout 60 tools/microbench/bench_trsm.bin $args $mode 2>&1 | grep -E "check|tasks" | sed -n '1p;$p' | tee -a $OUT
  done
done
