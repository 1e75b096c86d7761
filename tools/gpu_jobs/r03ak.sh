#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s %-32s %.2f ms  residual %.2e' % ('$W', '$*', d['ms_per_step'], d['residual']))"
}
{
W=fem27; S=3
run PANGULU_HIP_HEAVY_FIRST=2
run PANGULU_HIP_HEAVY_FIRST=1
run PANGULU_HIP_HEAVY_FIRST=0
run PANGULU_HIP_HEAVY_FIRST=2
W=shell; S=20
run PANGULU_HIP_HEAVY_FIRST=2
run PANGULU_HIP_HEAVY_FIRST=1
run PANGULU_HIP_HEAVY_FIRST=0
} 2>&1 | tee gpurun_out/r03ak_heavy_first.log
