#!/bin/bash
# separators in k-d order (same fill, fuller 16 x 16 pieces): parity slice, then A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py -m gpu -q -x -k "midsize or factors_match or dense_paths" ) 2>&1 | tail -2
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s %-44s %.2f ms  %.0f GFLOP/s  residual %.2e  factor_check %.2e' % ('$W', '$*', d['ms_per_step'], d['value'], d['residual'], d.get('factor_check', -1)))"
}
{
W=fem27; S=3
run PANGULU_AMD_SEPARATOR_ORDER=kd
run PANGULU_AMD_SEPARATOR_ORDER=natural
W=shell; S=20
run PANGULU_AMD_SEPARATOR_ORDER=kd
run PANGULU_AMD_SEPARATOR_ORDER=natural
W=poisson; S=5
run PANGULU_AMD_SEPARATOR_ORDER=kd
run PANGULU_AMD_SEPARATOR_ORDER=natural
} 2>&1 | tee gpurun_out/r03an_separator_order.log
