#!/bin/bash
# round 5: deferral depth with a "near its panel task" exemption (PANGULU_AMD_LOOKAHEAD_NEAR_LEVELS), headline workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05near}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
CASES=${CASES:-3:0 5:8 5:64 8:64 8:512}
for c in $CASES; do
  set -- ${c/:/ }
  run minq$1_near$2 PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$1 PANGULU_AMD_LOOKAHEAD_NEAR_LEVELS=$2 -- --steps 4 --warmup 1
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
