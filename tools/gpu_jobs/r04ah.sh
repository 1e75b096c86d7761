#!/bin/bash
# the default workload WITHOUT coordinates (what a SuiteSparse file gives the solver): multilevel nested dissection on the graph alone
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time PANGULU_AMD_TRACE_INIT=1 timeout 1500 python bench.py --no-coords --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sched-steps --no-profile-pass ) > gpurun_out/r04ah_elastic3d_77_nocoords.log 2> gpurun_out/r04ah_elastic3d_77_nocoords.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04ah_elastic3d_77_nocoords.log').readline())
c = d['config']
print(c['workload'][:80], 'F %.3e' % c['flop'], d['ms_per_step'], d['value'], d['residual'], d['factor_check'], 'init', d['init_s'], d.get('hbm_breakdown_GB'))
PY
tail -5 gpurun_out/r04ah_elastic3d_77_nocoords.err | cut -c1-200
