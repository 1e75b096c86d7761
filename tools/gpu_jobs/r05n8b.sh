#!/bin/bash
# round 5: eight ranks sharing the one GPU, deferral of shallow queues off (round 4's behaviour) for comparison with r05n8.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05n8b}
( PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=1 timeout 900 python bench.py --gpus 8 --transport ipc --workload elastic3d --size 56 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > gpurun_out/${TAG}_elastic3d_56_n8_minq1.json.log 2> gpurun_out/${TAG}_elastic3d_56_n8_minq1.err
echo "rc $?"
( timeout 900 python bench.py --gpus 8 --transport ipc --workload elastic3d --size 56 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > gpurun_out/${TAG}_elastic3d_56_n8_minq3.json.log 2> gpurun_out/${TAG}_elastic3d_56_n8_minq3.err
echo "rc $?"
( timeout 900 python bench.py --gpus 2 --transport ipc --workload elastic3d --size 56 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > gpurun_out/${TAG}_elastic3d_56_n2_minq3.json.log 2> gpurun_out/${TAG}_elastic3d_56_n2_minq3.err
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
