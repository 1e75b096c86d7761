#!/bin/bash
# round 4: where pangulu_init spends its time on the default bench matrix (after the symbolic phase's rewrite)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
PANGULU_AMD_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary --no-sched-steps 2>&1 | grep -E "init:|symbolic:|preprocess:|metric" | cut -c1-250 > gpurun_out/r04r_init_breakdown.log
cat gpurun_out/r04r_init_breakdown.log
