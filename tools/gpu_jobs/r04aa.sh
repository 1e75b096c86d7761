#!/bin/bash
# A/B: the dense-front launch on a stream of its own beside the general launch (PANGULU_HIP_FRONT_FORK)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04aa_front_fork_ab.log
: > $OUT
for w in "elastic3d --size 48" "fem27 --size 80" "fem27 --size 96"; do
for ff in 0 1 0 1; do
  line=$(PANGULU_HIP_FRONT_FORK=$ff timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary --no-sched-steps 2>/dev/null | tail -1)
  echo "FRONT_FORK=$ff $w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); c=d['config']; print('%.2f ms  %.2f TFLOP/s  residual %.2e  factor check %.2e' % (d['ms_per_step'], d['value']/1e3, d['residual'], d['factor_check']))" "$line" 2>&1 | tail -1)" | tee -a $OUT
done; done
