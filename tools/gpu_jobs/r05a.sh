#!/bin/bash
# round 5, first contact of the pieces kernel (pg_hip_pieces.h): stand-alone check + timing against the tilesv kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r05a_pieces_standalone.log
: > $OUT
B=tools/microbench/front_gemm.bin
for args in "40 1 45" "40 4 45" "40 1 30" "40 1 70"; do
  echo "=== front_gemm $args ===" | tee -a $OUT
  timeout 300 $B $args 2>&1 | tee -a $OUT | grep -v "^check" | tail -12
done
grep "^check" $OUT | sort | uniq -c | tee gpurun_out/r05a_checks.txt
for k in 1 2 3 4 5 6 8; do
  echo "=== k = $k ===" | tee -a $OUT
  timeout 300 $B 40 1 -$k 2>&1 | grep "^time" | tee -a $OUT
done
