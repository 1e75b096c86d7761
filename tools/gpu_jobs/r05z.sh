#!/bin/bash
# round 5: launch log of the profiled factorisation on the headline workload (update launches by size)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05z}
PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/${TAG}_launch_log.txt timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps --steps 2 --warmup 1 > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
python tools/launch_log_summary.py gpurun_out/${TAG}_launch_log.txt | tee gpurun_out/${TAG}_launch_log_summary.txt
PANGULU_AMD_LOOKAHEAD_MIN_QUEUE=1 PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/${TAG}_launch_log_minq1.txt timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps --steps 2 --warmup 1 > gpurun_out/${TAG}_bench_minq1.json.log 2> gpurun_out/${TAG}_bench_minq1.err
python tools/launch_log_summary.py gpurun_out/${TAG}_launch_log_minq1.txt | tee gpurun_out/${TAG}_launch_log_minq1_summary.txt
gzip -f gpurun_out/${TAG}_launch_log.txt gpurun_out/${TAG}_launch_log_minq1.txt
