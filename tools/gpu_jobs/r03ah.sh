#!/bin/bash
# the default bench line of the committed tree (roofline.traffic from profiles/hbm_traffic.json of this build) + smoke
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python bench.py ) > gpurun_out/r03ah_bench_default.log 2> gpurun_out/r03ah_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03ah_bench_default.log').readline())
r = d['roofline']
print(d['ms_per_step'], d['value'], d['residual'], d['factor_check'])
print({k: r.get(k) for k in ('achieved', 'frac', 'traffic', 'traffic_over_algorithmic', 'traffic_unit', 'traffic_note')})
print(d['cpu_baseline']['value'], d['config'].get('workload'))
PY
tail -3 gpurun_out/r03ah_bench_default.err
