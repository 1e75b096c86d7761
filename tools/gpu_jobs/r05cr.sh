#!/bin/bash
# round 5: the complex class (CR64, nb = 128) on the final build, deferral of shallow queues on (default) and off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05cr}
OUT=gpurun_out/${TAG}_cr64.log
: > $OUT
for N in 64 80; do
  for q in 3 1; do
    echo "== poisson3d($N) CR64 nb 128, PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q" | tee -a $OUT
    PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q timeout 900 python tools/bench_cr64.py $N 128 2>&1 | grep -v amdgpu.ids | head -1 | tee -a $OUT
  done
done
