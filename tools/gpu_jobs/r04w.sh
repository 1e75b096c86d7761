#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04w_unfixed_candidates.log
: > $OUT
run() { echo "== $*" | tee -a $OUT; env PYTHONPATH=$R "${@:2}" timeout 600 python tests/env_switch_worker.py $1 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT; }
for m in shell fem27; do
for c in 8 64; do
for pm in 50 100 300 600; do
run $m PANGULU_HIP_LAUNCH_CHUNK=$c PANGULU_HIP_DENSE_PERMILLE=$pm
done; done; done
