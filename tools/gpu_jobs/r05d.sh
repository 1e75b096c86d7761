#!/bin/bash
# round 5: cycle probes inside the tilesv and the pieces kernel (tools/microbench/front_gemm.hip -DTL_PROBE)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05d}_probe.log
: > $OUT
B=tools/microbench/front_gemm_probe.bin
for args in "40 4 -8" "40 4 -4" "40 4 45" "40 1 45" "40 2 30"; do
  echo "=== front_gemm(probe) $args ===" | tee -a $OUT
  timeout 300 $B $args 2>&1 | grep "^time\|^probe\|^item" | tee -a $OUT
done
