#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity_scale.py -m gpu -q -x -k "dense_paths" ) > gpurun_out/r03v_pytest.log 2>&1
tail -5 gpurun_out/r03v_pytest.log
timeout 1200 python tools/bench_types.py 48 r64 r32 2>&1 | grep -E "poisson3d|launches|Error|error" | tee gpurun_out/r03v_bench_types.log
timeout 600 python bench.py --gpu-worker --workload poisson3d --size 48 --steps 5 --warmup 2 2>/dev/null | grep '"metric"' | cut -c1-400
