#!/bin/bash
# round 5: eight ranks sharing the one GPU on a mid-size matrix (deferral of shallow queues active: > 8192 queued updates per rank), replay on and off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05n8}
( timeout 900 python bench.py --gpus 8 --workload elastic3d --size 56 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > gpurun_out/${TAG}_elastic3d_56_n8.json.log 2> gpurun_out/${TAG}_elastic3d_56_n8.err
echo "rc $?"; tail -3 gpurun_out/${TAG}_elastic3d_56_n8.err
( timeout 900 python bench.py --gpus 8 --workload elastic3d --size 56 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-multi-replay ) > gpurun_out/${TAG}_elastic3d_56_n8_noreplay.json.log 2> gpurun_out/${TAG}_elastic3d_56_n8_noreplay.err
echo "rc $?"
( timeout 900 python bench.py --gpus 1 --workload elastic3d --size 56 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > gpurun_out/${TAG}_elastic3d_56_n1.json.log 2> gpurun_out/${TAG}_elastic3d_56_n1.err
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
