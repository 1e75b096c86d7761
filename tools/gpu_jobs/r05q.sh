#!/bin/bash
# round 5: lone dense panel solves against the depth of the factor-tile queue (tools/microbench/bench_trsm.hip, -DTRSM_SETS)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05q}_trsm_sets.log
: > $OUT
for s in 2 3 4 5; do
  for args in "1 1 1" "1 0 1" "4 1 1" "64 1 4" "1024 1 64"; do
    echo "== sets $s args $args" | tee -a $OUT
    timeout 60 tools/microbench/bench_trsm_s$s.bin $args 2>&1 | tail -1 | tee -a $OUT
  done
done
