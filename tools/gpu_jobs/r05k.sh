#!/bin/bash
# round 5: multi-rank replay as the default (tests), then the driver's command line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05k}
( timeout 1800 python -m pytest tests/test_multirank.py tests/test_gpu_smoke_bench.py tests/test_update_values.py -m gpu -q --durations=8 ) > gpurun_out/${TAG}_multirank_tests.log 2>&1
tail -15 gpurun_out/${TAG}_multirank_tests.log
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_default.json.log 2> gpurun_out/${TAG}_bench_default.err
echo "bench rc $?"
python tools/ab_summary.py gpurun_out/${TAG}_bench_default.json.log
