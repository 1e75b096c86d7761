#!/bin/bash
# thresholds re-checked on the k-d ordered pattern (more tiles are completely live now)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload fem27 --size 112 --steps 3 --warmup 1 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-60s %.1f ms  residual %.2e' % ('$*', d['ms_per_step'], d['residual']))"
}
{
run PANGULU_AMD_X=0
run PANGULU_HIP_FRONT_MIN_WGS=1024
run PANGULU_HIP_FRONT_MIN_WGS=4096
run PANGULU_HIP_FRONT_MIN_WGS=32768
run PANGULU_HIP_FRONT_UNIT=2
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=16
} 2>&1 | tee gpurun_out/r03aq_threshold_recheck_kd.log
