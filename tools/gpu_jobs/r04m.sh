#!/bin/bash
# round 4: block order 128 against 256 on the latency-bound ldoor-class matrix (and the headline classes for reference)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04m_nb128.log
: > $OUT
for w in "shell" "fem27 --size 80" "poisson --size 80"; do for nb in 256 128; do
  line=$(timeout 900 python bench.py --workload $w --nb $nb --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary 2>/dev/null | tail -1)
  echo "$w nb=$nb :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f ms  %.2f TFLOP/s  F %.3e  residual %.2e  init %.1f s  hbm %.0f GB' % (d['ms_per_step'], d['value']/1e3, d['config']['flop'], d['residual'], d['init_s'], d['hbm_used_GB']))" "$line")" | tee -a $OUT
done; done
