#!/bin/bash
# round 4: the dense-front kernel got faster (destination preloaded): from how many all-live workgroups should a launch hand them to it?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04i_front_min_wgs.log
: > $OUT
for w in 8192 2048 512 64; do
  line=$(PANGULU_HIP_FRONT_MIN_WGS=$w timeout 900 python bench.py --workload fem27 --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary 2>/dev/null | tail -1)
  echo "fem27(112) FRONT_MIN_WGS=$w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f ms  %.2f TFLOP/s  residual %.2e' % (d['ms_per_step'], d['value']/1e3, d['residual']))" "$line")" | tee -a $OUT
done
for w in 8192 512; do
  line=$(PANGULU_HIP_FRONT_MIN_WGS=$w timeout 900 python bench.py --workload shell --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary 2>/dev/null | tail -1)
  echo "shell(398) FRONT_MIN_WGS=$w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f ms  %.2f TFLOP/s  residual %.2e' % (d['ms_per_step'], d['value']/1e3, d['residual']))" "$line")" | tee -a $OUT
done
