#!/bin/bash
# round 4: how much of a replayed factorisation is the host issuing launches?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for w in "shell" "poisson" "fem27 --size 64"; do
  PANGULU_AMD_TRACE=1 timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary 2>&1 | grep -E "replay:|metric" | cut -c1-200 | tail -4
done > gpurun_out/r04l_replay_issue_time.log 2>&1
cat gpurun_out/r04l_replay_issue_time.log
