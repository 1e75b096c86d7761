#!/bin/bash
# round 5: ring kernel for dense TSTRF: stand-alone check + timing, the solve-related GPU tests, A/B in the factorisation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05ring}
bash tools/gpu_jobs/r05q.sh
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_operators.py -m gpu -q -x -k "not cr64 and not cr32 and not complex" --durations=5 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-sched-steps "$@" > gpurun_out/${TAG}_$name.json.log 2> gpurun_out/${TAG}_$name.err; tail -2 gpurun_out/${TAG}_$name.err; }
for r in 0 1; do
  run shell398_ring$r PANGULU_HIP_TRSM_RING=$r -- --workload shell --size 398 398 --steps 10 --warmup 3
  run fem27_64_ring$r PANGULU_HIP_TRSM_RING=$r -- --workload fem27 --size 64 --steps 5 --warmup 2
  run elastic3d_48_ring$r PANGULU_HIP_TRSM_RING=$r -- --workload elastic3d --size 48 --steps 5 --warmup 2
done
python tools/ab_summary.py gpurun_out/${TAG}_*.json.log | tee gpurun_out/${TAG}_summary.txt
