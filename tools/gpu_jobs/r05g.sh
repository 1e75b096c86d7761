#!/bin/bash
# round 5: XCD unit of the update launches inside the factorisation (PANGULU_HIP_TILES_UNIT / PANGULU_HIP_FRONT_UNIT), one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05g}
run() { # name, env..., -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err
}
for W in "elastic3d 48" "fem27 64"; do
  set -- $W
  run ${1}_${2}_default A=1 -- --workload $1 --size $2 --steps 5 --warmup 2
  run ${1}_${2}_tiles_unit8 PANGULU_HIP_TILES_UNIT=8 -- --workload $1 --size $2 --steps 5 --warmup 2
  run ${1}_${2}_tiles_unit4 PANGULU_HIP_TILES_UNIT=4 -- --workload $1 --size $2 --steps 5 --warmup 2
  run ${1}_${2}_front_unit4 PANGULU_HIP_FRONT_UNIT=4 -- --workload $1 --size $2 --steps 5 --warmup 2
  run ${1}_${2}_both8 PANGULU_HIP_TILES_UNIT=8 PANGULU_HIP_FRONT_UNIT=8 -- --workload $1 --size $2 --steps 5 --warmup 2
  run ${1}_${2}_default_again A=1 -- --workload $1 --size $2 --steps 5 --warmup 2
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
