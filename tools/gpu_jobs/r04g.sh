#!/bin/bash
# round 4: (1) dense-front kernel with the destination preloaded into the accumulators, stand-alone A/B; (2) parity of the update kernels;
# (3) elastic3d(77): Serena's n AND row length (3 dofs x 15-point node stencil, 61 M entries) on one GPU; (4) fem27(112) default line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( cd tools/microbench && for q in 1 2 4; do echo "== preload, Q=$q"; timeout 300 ./front_gemm.bin 40 $q 100 | grep -E "time|check.*front kernel, 2"; echo "== no preload, Q=$q"; timeout 300 ./front_gemm_nopreload.bin 40 $q 100 | grep -E "time"; done ) > gpurun_out/r04g_front_preload.log 2>&1
grep -E "==|front kernel, 2 LDS" gpurun_out/r04g_front_preload.log | cut -c1-170
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/r04g_parity.log 2>&1
tail -3 gpurun_out/r04g_parity.log
( time PANGULU_AMD_TRACE=1 timeout 1500 python bench.py --workload elastic3d --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > gpurun_out/r04g_bench_elastic3d_77.log 2> gpurun_out/r04g_bench_elastic3d_77.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04g_bench_elastic3d_77.log').readline())
    print("elastic3d(77):", d['ms_per_step'], d['value'], d['residual'], d['factor_check'], d['config']['n'], d['config']['nnz'], d['config']['flop'], d['config']['symbolic_nnz'], d['hbm_used_GB'], d.get('hbm_breakdown_GB'), d['init_s'], d['gstrs_s'], d['roofline']['frac'] if d.get('roofline') else None)
    print({k:(v['ms'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e:
    print("elastic3d FAILED", e)
PY
tail -3 gpurun_out/r04g_bench_elastic3d_77.err | cut -c1-300
( time timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary ) > gpurun_out/r04g_bench_default.log 2> gpurun_out/r04g_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04g_bench_default.log').readline())
print("fem27(112):", d['ms_per_step'], d['value'], d['residual'], d['roofline']['frac'], {k:(v['ms'],v['launches']) for k,v in d['kernels'].items()})
PY
