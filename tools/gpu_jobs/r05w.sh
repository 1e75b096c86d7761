#!/bin/bash
# round 5: deferral of shallow update queues as the default (MIN_QUEUE 3 from DEFER_FROM queued updates on): A/B on small and large workloads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05w}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
for q in 1 3; do
  run shell398_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload shell --size 398 398 --steps 10 --warmup 3
  run fem27_64_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload fem27 --size 64 --steps 5 --warmup 2
  run elastic3d_48_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload elastic3d --size 48 --steps 5 --warmup 2
  run fem27_112_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload fem27 --size 112 --steps 4 --warmup 1
  run kkt120_minq$q PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=$q -- --workload kkt --size 120 --steps 4 --warmup 1
done
run elastic3d_77_default -- --steps 4 --warmup 1
run elastic3d_77_from0 PANGULU_AMD_LOOKAHEAD_DEFER_FROM=0 -- --steps 4 --warmup 1
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
