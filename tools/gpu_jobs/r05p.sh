#!/bin/bash
# round 5: the whole GPU suite on the final build, then the driver's bench command line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05p}
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/${TAG}_gputests.log 2>&1
tail -22 gpurun_out/${TAG}_gputests.log
( timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_default.json.log 2> gpurun_out/${TAG}_bench_default.err
echo "bench rc $?"
python tools/ab_summary.py gpurun_out/${TAG}_bench_default.json.log
