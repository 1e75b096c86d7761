#!/bin/bash
# update work items with the longest queues first (PANGULU_HIP_HEAVY_FIRST): parity slice, then A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py -m gpu -q -x -k "midsize or general_update or dense_front or factors_match" ) 2>&1 | tail -2
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s %-32s %.2f ms  residual %.2e' % ('$W', '$*', d['ms_per_step'], d['residual']))"
}
{
W=fem27; S=3
run PANGULU_HIP_HEAVY_FIRST=1
run PANGULU_HIP_HEAVY_FIRST=0
run PANGULU_HIP_HEAVY_FIRST=1
run PANGULU_HIP_HEAVY_FIRST=0
W=shell; S=20
run PANGULU_HIP_HEAVY_FIRST=1
run PANGULU_HIP_HEAVY_FIRST=0
} 2>&1 | tee gpurun_out/r03aj_heavy_first.log
