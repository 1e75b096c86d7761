#!/bin/bash
# profile recipe + default bench line of the final tree (the source hash now covers the host sources as well)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
bash tools/profile_recipe.sh r03ap_fem27_112 > gpurun_out/r03ap_profile_recipe.log 2>&1
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
( time timeout 1500 python bench.py ) > gpurun_out/r03ap_bench_default.log 2> gpurun_out/r03ap_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03ap_bench_default.log').readline())
r = d['roofline']
print(d['ms_per_step'], d['value'], d['residual'], d['factor_check'])
print({k: r.get(k) for k in ('achieved', 'frac', 'traffic', 'traffic_over_algorithmic', 'traffic_note')})
PY
head -6 gpurun_out/r03ap_fem27_112_table.md | cut -c1-200
