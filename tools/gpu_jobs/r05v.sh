#!/bin/bash
# round 5: look-ahead calls with a floor on their size (PANGULU_AMD_LOOKAHEAD_MIN_TASKS) at MIN_QUEUE = 4 / 8 on the headline workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05v}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
for c in ${CASES:-"4 2048" "4 8192" "8 4096" "8 16384"}; do
  set -- ${c/:/ }
  run elastic3d_77_minq$1_mint$2 PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$1 PANGULU_AMD_LOOKAHEAD_MIN_TASKS=$2 -- --steps 4 --warmup 1
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
