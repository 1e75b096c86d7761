#!/bin/bash
# round 3, tenth GPU job: replays read their descriptors from HBM; tests of the static schedule
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_update_values.py tests/test_gpu_smoke_bench.py tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/r03j_pytest.log 2>&1; tail -6 gpurun_out/r03j_pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-profile-pass"
run() { name=$1; shift
  envs=""; while [ $# -gt 0 ] && [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done
  env $envs $B "$@" > gpurun_out/r03j_$name.log 2>&1
  grep -a '"metric"' gpurun_out/r03j_$name.log | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('$name: ms_per_step %.2f %s residual %.2e factor_check %.2e replayed %s hbm %.1f GB' % (l['ms_per_step'], l['step_ms'], l['residual'], l['factor_check'], l.get('static_schedule_replayed'), l['hbm_used_GB']))"
}
run fem112 --steps 5 --warmup 2
run shell --workload shell --steps 20 --warmup 3
run fem112_lazy PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 --steps 5 --warmup 2
run fem112_bg PANGULU_HIP_BACKGROUND_UPDATES=1 PANGULU_AMD_PANEL_FIRST=1 --steps 5 --warmup 2
run shell_bg PANGULU_HIP_BACKGROUND_UPDATES=1 PANGULU_AMD_PANEL_FIRST=1 --workload shell --steps 20 --warmup 3
run shell_lazy PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 --workload shell --steps 20 --warmup 3
run fem112_front0 PANGULU_HIP_FRONT_STAGES=0 --steps 5 --warmup 2
