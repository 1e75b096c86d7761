#!/bin/bash
# round 4: N ranks SHARING the one GPU of the box (time-sliced: not a scaling number) -- the scheduler in the loop against the replay of the ranks' logs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04p_multi_replay_shared_gpu.log
: > $OUT
for n in 2 4; do for w in "shell" "fem27 --size 64"; do for mode in "" "--multi-replay"; do
  line=$(PANGULU_AMD_TRACE=0 timeout 900 python bench.py --gpus $n --workload $w --transport ipc --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass $mode 2>gpurun_out/r04p_last.err | tail -1)
  echo "$n ranks on one GPU, $w, ${mode:-scheduler} :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f ms  residual %.2e  factor check %.2e  transport %s  replayed %s  steps %s' % (d['ms_per_step'], d['residual'], d['factor_check'], d['config']['transport'], d['static_schedule_replayed'], d['step_ms']))" "$line" 2>&1 | tail -1)" | tee -a $OUT
done; done; done
tail -5 gpurun_out/r04p_last.err | cut -c1-200
