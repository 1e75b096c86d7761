#!/bin/bash
# round 5: the pieces kernel inside the factorisation, A/B against the tilesv kernel on one box (PANGULU_HIP_TILES_STAGES=5 / 2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05b}
run() { # name, env..., -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err
  tail -3 gpurun_out/${TAG}_$name.err
}
for W in "elastic3d 40" "fem27 56"; do
  set -- $W
  run ${1}_${2}_stages2 PANGULU_HIP_TILES_STAGES=2 -- --workload $1 --size $2 --steps 5 --warmup 2
  run ${1}_${2}_stages5 PANGULU_HIP_TILES_STAGES=5 -- --workload $1 --size $2 --steps 5 --warmup 2
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
