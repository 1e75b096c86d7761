#!/bin/bash
# round 5: the register-resident GETRF (pg_hip_getrf_pipe.h): parity cases, then A/B on the ldoor-class matrix
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05l}
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_operators.py -m gpu -q -x -k "not cr64 and not cr32 and not complex" --durations=5 ) > gpurun_out/${TAG}_getrf_tests.log 2>&1
tail -15 gpurun_out/${TAG}_getrf_tests.log
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
run shell398_pipe0 PANGULU_HIP_GETRF_PIPE=0 -- --workload shell --size 398 398 --steps 10 --warmup 3
run shell398_pipe1 PANGULU_HIP_GETRF_PIPE=1 -- --workload shell --size 398 398 --steps 10 --warmup 3
run fem27_64_pipe0 PANGULU_HIP_GETRF_PIPE=0 -- --workload fem27 --size 64 --steps 5 --warmup 2
run fem27_64_pipe1 PANGULU_HIP_GETRF_PIPE=1 -- --workload fem27 --size 64 --steps 5 --warmup 2
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
