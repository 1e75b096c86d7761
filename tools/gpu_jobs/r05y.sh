#!/bin/bash
# round 5: neighbours of the scheduler's look-ahead settings with the deferral of shallow queues in place (headline workload)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05y}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
run base X=1 -- --steps 4 --warmup 1
run maxgetrf16 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=16 -- --steps 4 --warmup 1
run maxgetrf64 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=64 -- --steps 4 --warmup 1
run frontmin4096 PANGULU_HIP_FRONT_MIN_WGS=4096 -- --steps 4 --warmup 1
run frontmin16384 PANGULU_HIP_FRONT_MIN_WGS=16384 -- --steps 4 --warmup 1
run deferfrom65536 PANGULU_AMD_LOOKAHEAD_DEFER_FROM=65536 -- --steps 4 --warmup 1
run panelfirst0 PANGULU_AMD_PANEL_FIRST=0 -- --steps 4 --warmup 1
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
