#!/bin/bash
# R32 / CR32 on double mirrors through the f64 dense kernels: value-type parity tests, then speed per type with the dense paths on and off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_parity.py tests/test_gpu_operators.py -m gpu -q -x -k "type or r32 or cr32 or cr64 or other or operator or dense_paths" ) > gpurun_out/r03u_pytest.log 2>&1
tail -8 gpurun_out/r03u_pytest.log
timeout 1200 python tools/bench_types.py 48 2>&1 | grep -E "poisson3d|Error|error" | tee gpurun_out/r03u_bench_types.log
