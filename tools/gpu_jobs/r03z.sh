#!/bin/bash
# GETRF -> dense-solve chase: parity, then A/B on one box (PANGULU_HIP_CHASE=0 holds nothing)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_parity_scale.py tests/test_update_values.py -m gpu -q -x -k "midsize or kkt or recorded or dense_paths" ) > gpurun_out/r03z_pytest.log 2>&1
tail -6 gpurun_out/r03z_pytest.log
run() {
  env "$@" timeout 600 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['kernels']
print('%-8s %-28s %.2f ms  residual %.2e  factor_check %.2e  solves %.2f getrf %.2f ms' % ('$W', '$*', d['ms_per_step'], d['residual'], d.get('factor_check', -1), k['tstrf']['ms'] + k.get('gessm', {}).get('ms', 0.0), k['getrf']['ms']))"
}
{
W=shell; S=20
run PANGULU_HIP_CHASE=1
run PANGULU_HIP_CHASE=0
run PANGULU_HIP_CHASE=1
run PANGULU_HIP_CHASE=0
W=fem27; S=3
run PANGULU_HIP_CHASE=1
run PANGULU_HIP_CHASE=0
} 2>&1 | tee gpurun_out/r03z_chase_ab.log
