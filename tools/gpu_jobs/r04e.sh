#!/bin/bash
# round 4: the default line with the secondary workload fixed; a Serena-SCALE matrix (fem27(128): n = 2.1 M, 56 M entries, F ~ 6e13) on one GPU
# with the records' snapshot on the host; the new test cases
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
free -g | head -2; nproc
( time timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline ) > gpurun_out/r04e_bench_default.log 2> gpurun_out/r04e_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04e_bench_default.log').readline())
print(d['ms_per_step'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'), d['residual'], d.get('hbm_breakdown_GB'))
print([(s['workload'][:12], s['ms_per_step'], s['value'], s['residual']) for s in d.get('secondary') or []])
PY
avail=$(free -g | awk '/Mem:/ {print $7}')
if [ "$avail" -gt 450 ]; then
  ( time PANGULU_HIP_MIRROR_FRACTION=0.8 PANGULU_AMD_TRACE=1 timeout 1500 python bench.py --workload fem27 --size 128 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary ) > gpurun_out/r04e_bench_fem27_128.log 2> gpurun_out/r04e_bench_fem27_128.err
  python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04e_bench_fem27_128.log').readline())
    print("fem27(128):", d['ms_per_step'], d['value'], d['residual'], d['factor_check'], d['config']['n'], d['config']['nnz'], d['config']['flop'], d['hbm_used_GB'], d.get('hbm_breakdown_GB'), d['init_s'])
except Exception as e:
    print("fem27(128) FAILED", e)
PY
  tail -5 gpurun_out/r04e_bench_fem27_128.err | cut -c1-300
else
  echo "host has only ${avail} GB available: Serena-scale run skipped"
fi
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "switch or snapshot or smoke or bench" ) > gpurun_out/r04e_gputests_subset.log 2>&1
tail -4 gpurun_out/r04e_gputests_subset.log
