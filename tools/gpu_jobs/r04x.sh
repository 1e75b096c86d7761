#!/bin/bash
# round 4, after the fix of the background-stream ordering between the launches of one call: the whole GPU suite, then the complex
# workloads of r04t again (their larger sizes ran with a non-default dense threshold there and hit the bug)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 2700 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04x_gputests.log 2>&1
tail -4 gpurun_out/r04x_gputests.log
OUT=gpurun_out/r04x_cr64.log
: > $OUT
for N in 64 80 96; do timeout 900 python tools/bench_cr64.py $N 128 2>&1 | tail -3 | tee -a $OUT; done
timeout 900 python tools/bench_cr64.py 80 256 2>&1 | tail -3 | tee -a $OUT
