#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_parity.py tests/test_gpu_operators.py -m gpu -q -x -k "cr64 or cr32 or other_value or dense_paths or complex" ) > gpurun_out/r03ad_pytest.log 2>&1
tail -5 gpurun_out/r03ad_pytest.log
bash tools/gpu_jobs/r03ac.sh 2>&1 | grep -E "poisson|calls" | head -12
