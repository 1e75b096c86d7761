#!/bin/bash
# round 5: the tree as committed last: smoke, bench entry points, switch list
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05last}
( timeout 1500 python -m pytest tests/test_gpu_smoke_bench.py tests/test_gpu_operators.py tests/test_gpu_parity.py -m gpu -q -x --durations=5 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
( timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline ) > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
echo "bench rc $?"
python tools/ab_summary.py gpurun_out/${TAG}_bench.json.log
