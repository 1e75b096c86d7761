#!/bin/bash
# round 5: PANGULU_AMD_LOOKAHEAD_MIN_QUEUE on the headline workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05u}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
for q in ${QUEUES:-1 2 4 16 1000}; do
  run elastic3d_77_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --steps 4 --warmup 1
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
