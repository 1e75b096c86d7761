#!/bin/bash
# round 5: look-ahead calls that leave shallow update queues alone (PANGULU_AMD_LOOKAHEAD_MIN_QUEUE) against the default and the lazy mode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05t}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
for q in ${QUEUES:-1 2 4 8 1000}; do
  run elastic3d_48_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload elastic3d --size 48 --steps 5 --warmup 2
  run fem27_64_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload fem27 --size 64 --steps 5 --warmup 2
done
run elastic3d_48_lazy PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 -- --workload elastic3d --size 48 --steps 5 --warmup 2
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
