#!/bin/bash
# nlpkkt class (BASELINE configs[3]: n = 3.54 M) at its full order on one GPU: kkt(96), kkt(120) (n = 3.46 M), with and without coordinates
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04am_kkt_class.log
: > $OUT
for w in "kkt --size 96" "kkt --size 120" "kkt --size 120 --no-coords"; do
  line=$(timeout 1200 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary --no-sched-steps 2>gpurun_out/r04am_err.log | tail -1)
  echo "R64 $w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); c=d['config']; print('n %d  nnz %d  F %.3e  %.2f ms  %.2f TFLOP/s  residual %.2e  factor check %.2e  gstrs %.3f s  init %.1f s  hbm %s' % (c['n'], c['nnz'], c['flop'], d['ms_per_step'], d['value']/1e3, d['residual'], d['factor_check'], d['gstrs_s'], d['init_s'], d.get('hbm_used_GB')))" "$line" 2>&1 | tail -1)" | tee -a $OUT
done
tail -3 gpurun_out/r04am_err.log | cut -c1-200
