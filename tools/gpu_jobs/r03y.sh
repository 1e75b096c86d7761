#!/bin/bash
# dense solves with factor tiles requested two stages ahead (default build) against one stage (tools/ab_lib): parity, then A/B on one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( true || time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py -m gpu -q -x -k "midsize or kkt or thresholds or factors_match" ) > gpurun_out/r03y_pytest.log 2>&1
tail -4 gpurun_out/r03y_pytest.log
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['kernels']
print('%-12s %-40s %.2f ms  residual %.2e  solves %.2f getrf %.2f ms' % ('$W', '$*', d['ms_per_step'], d['residual'], k['tstrf']['ms'] + k.get('gessm', {}).get('ms', 0.0), k['getrf']['ms']))"
}
{
W=shell; S=20
run PANGULU_AMD_X=0
run PANGULU_AMD_LIB_DIR=$R/tools/ab_lib
run PANGULU_AMD_X=0
run PANGULU_AMD_LIB_DIR=$R/tools/ab_lib
W=fem27; S=3
run PANGULU_AMD_X=0
run PANGULU_AMD_LIB_DIR=$R/tools/ab_lib
} 2>&1 | tee gpurun_out/r03y_trsm_prefetch_ab.log
