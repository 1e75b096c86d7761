#!/bin/bash
# round 4, first box: the whole GPU suite (new: default-map multi-rank cases at 2/4/8 ranks) + the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04a_gputests.log 2>&1
tail -5 gpurun_out/r04a_gputests.log
( time timeout 900 python bench.py --steps 6 --warmup 2 ) > gpurun_out/r04a_bench_default.log 2> gpurun_out/r04a_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04a_bench_default.log').readline())
print(d['ms_per_step'], d['value'], d['residual'], d['factor_check'])
PY
