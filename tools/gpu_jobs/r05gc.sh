#!/bin/bash
# round 5: queue chunks (updates of one destination cut into concurrent chunks with atomics) against the deferral depth, headline workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05gc}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
CASES=${CASES:-8:3 16:3 32:3 32:6 0:4}
for c in $CASES; do
  set -- ${c/:/ }
  run chunk$1_minq$2 PANGULU_HIP_GROUP_CHUNK=$1 PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$2 -- --steps 4 --warmup 1
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
