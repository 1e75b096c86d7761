#!/bin/bash
# other workload classes with the final build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  timeout 900 python bench.py --gpu-worker --workload $1 --size $2 --steps 4 --warmup 2 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s %-6s %.1f ms  %.0f GFLOP/s  residual %.2e  factor_check %.2e  T*/t %.2f' % ('$1', '$2', d['ms_per_step'], d['value'], d['residual'], (d.get('factor_check') or -1), (d.get('model') or {}).get('T_star_over_t_gstrf', -1)))"
}
{
run poisson 96
run fem27 80
run fem27 64
run poisson 64
} 2>&1 | tee gpurun_out/r03ar_other_workloads.log
