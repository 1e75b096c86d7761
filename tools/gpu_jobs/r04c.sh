#!/bin/bash
# round 4: chunked triangular-solve kernels -- GPU suite, then the default line (gstrs_s, ms_per_step_scheduler_in_loop) and the old kernels for comparison
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04c_gputests.log 2>&1
tail -5 gpurun_out/r04c_gputests.log
( time timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/r04c_bench_default.log 2> gpurun_out/r04c_bench_default.err
( time PANGULU_HIP_SOLVE_CHUNKED=0 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass ) > gpurun_out/r04c_bench_oldsolve.log 2> gpurun_out/r04c_bench_oldsolve.err
python - <<'PY'
import json
for f in ("gpurun_out/r04c_bench_default.log", "gpurun_out/r04c_bench_oldsolve.log"):
    try:
        d = json.loads(open(f).readline())
        print(f, d['ms_per_step'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'), d['residual'], d['factor_check'], d.get('hbm_used_GB'))
    except Exception as e:
        print(f, "FAILED", e)
PY
