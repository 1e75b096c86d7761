#!/bin/bash
# last build of round 4: the other workload classes (parity cases, not bench lines), R64 through bench.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04ap_other_workloads.log
: > $OUT
for w in "poisson --size 96" "poisson --size 64" "fem27 --size 80" "fem27 --size 64" "kkt --size 40" "kkt --size 96" "elastic3d --size 48" "shell --size 300 300"; do
  line=$(timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary --no-sched-steps 2>/dev/null | tail -1)
  echo "R64 $w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); c=d['config']; print('n %d  nnz %d  F %.3e  %.2f ms  %.2f TFLOP/s  residual %.2e  factor check %.2e  gstrs %.3f s  init %.1f s' % (c['n'], c['nnz'], c['flop'], d['ms_per_step'], d['value']/1e3, d['residual'], d['factor_check'], d['gstrs_s'], d['init_s']))" "$line" 2>&1 | tail -1)" | tee -a $OUT
done
