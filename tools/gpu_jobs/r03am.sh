#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s %-40s %.2f ms  residual %.2e' % ('$W', '$*', d['ms_per_step'], d['residual']))"
}
{
W=fem27; S=3
run PANGULU_HIP_TILES_UNIT=1
run PANGULU_HIP_TILES_UNIT=2
run PANGULU_HIP_TILES_UNIT=8
run PANGULU_HIP_TILES_UNIT=1 PANGULU_HIP_FRONT_UNIT=4
run PANGULU_HIP_TILES_UNIT=1
} 2>&1 | tee gpurun_out/r03am_xcd_units.log
