#!/bin/bash
# round 5: pieces kernel v3 (scalar pipeline state, no tables): probes, stand-alone, in the factorisation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05f}
TAG=$TAG bash tools/gpu_jobs/r05d.sh > /dev/null
grep "^===\|^item\|^time" gpurun_out/${TAG}_probe.log | cut -c1-300
OUT=gpurun_out/${TAG}_standalone.log
: > $OUT
B=tools/microbench/front_gemm.bin
for args in "40 1 45" "40 4 45" "40 4 -2" "40 4 -4" "40 4 -8"; do
  echo "=== front_gemm $args ===" | tee -a $OUT
  timeout 300 $B $args 2>&1 | tee -a $OUT | grep "^time\|WRONG" | tail -4
done
grep -c "ok$" $OUT; grep "WRONG" $OUT
TAG=$TAG bash tools/gpu_jobs/r05b.sh
