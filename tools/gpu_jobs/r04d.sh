#!/bin/bash
# round 4: (1) experiment -- dense-front kernel with K = 32 per barrier (tools/experiments/front_k32.h) against the product kernels;
# (2) complex Poisson (BASELINE configs[4] class) at nb = 128 against nb = 256; (3) the default line with the secondary workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( cd tools/microbench && for q in 1 4; do timeout 300 ./front_gemm.bin 40 $q 100; done ) > gpurun_out/r04d_front_k32.log 2>&1
grep -E "check|time" gpurun_out/r04d_front_k32.log | cut -c1-200
for nb in 256 128; do for N in 48 64; do timeout 600 python tools/bench_cr64.py $N $nb 2>&1 | grep -v "=1001" ; done; done > gpurun_out/r04d_cr64_nb.log 2>&1
cat gpurun_out/r04d_cr64_nb.log | cut -c1-220
( time timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline ) > gpurun_out/r04d_bench_default.log 2> gpurun_out/r04d_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04d_bench_default.log').readline())
print(d['ms_per_step'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'), d['residual'], d['factor_check'], d.get('hbm_breakdown_GB'))
print(d.get('secondary'))
PY
tail -3 gpurun_out/r04d_bench_default.err
( time PANGULU_HIP_SOLVE_CHUNKED=0 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary ) > gpurun_out/r04d_bench_oldsolve.log 2> gpurun_out/r04d_bench_oldsolve.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04d_bench_oldsolve.log').readline())
print("old solve kernels:", d['ms_per_step'], d.get('gstrs_s'), d['residual'])
PY
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04d_gputests.log 2>&1
tail -4 gpurun_out/r04d_gputests.log
