#!/bin/bash
# round 4: first-touch densify jobs moved into a prologue on their own stream (replayed schedules) -- parity, then A/B on the bench matrices
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "parity or switch or update_values or smoke" ) > gpurun_out/r04k_gputests.log 2>&1
tail -4 gpurun_out/r04k_gputests.log
OUT=gpurun_out/r04k_early_densify.log
: > $OUT
run() { # label env -- args
  label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  line=$(env "${envs[@]}" PANGULU_AMD_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-profile-pass --no-secondary "$@" 2>gpurun_out/r04k_last.err | tail -1)
  echo "$label :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f ms  %.2f TFLOP/s  residual %.2e  factor check %.2e' % (d['ms_per_step'], d['value']/1e3, d['residual'], d['factor_check']))" "$line") $(grep -o 'schedule: .*' gpurun_out/r04k_last.err | head -1)" | tee -a $OUT
}
for e in 1 0; do
  run "shell(398) early_densify=$e" PANGULU_HIP_EARLY_DENSIFY=$e -- --workload shell --steps 10 --warmup 2
done
for e in 1 0; do
  run "fem27(112) early_densify=$e" PANGULU_HIP_EARLY_DENSIFY=$e -- --workload fem27 --steps 5 --warmup 2
done
for e in 1 0; do
  run "poisson3d(64) early_densify=$e" PANGULU_HIP_EARLY_DENSIFY=$e -- --workload poisson --steps 8 --warmup 2
done
run "shell(398) early_densify=1 again" PANGULU_HIP_EARLY_DENSIFY=1 -- --workload shell --steps 10 --warmup 2
