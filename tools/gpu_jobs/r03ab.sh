#!/bin/bash
# complex panels in the mirrors (zgetrf / ztrsm on two-plane mirrors): value-type parity tests, then speed per type
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_parity.py tests/test_gpu_operators.py -m gpu -q -x -k "cr64 or cr32 or other_value or dense_paths or complex" ) > gpurun_out/r03ab_pytest.log 2>&1
tail -12 gpurun_out/r03ab_pytest.log
timeout 1200 python tools/bench_types.py 48 cr64 cr32 2>&1 | grep -E "poisson3d|launches|Error|error" | cut -c1-330 | tee gpurun_out/r03ab_bench_types.log
PANGULU_HIP_COMPLEX_PANELS=0 timeout 600 python tools/bench_types.py 48 cr64 2>&1 | grep -E "poisson3d" | head -1 | cut -c1-330 | tee -a gpurun_out/r03ab_bench_types.log
