#!/bin/bash
# round 3, ninth GPU job: static schedule (record on the first gstrf, replay afterwards)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_update_values.py tests/test_gpu_smoke_bench.py -m gpu -x -q ) > gpurun_out/r03i_pytest.log 2>&1; tail -6 gpurun_out/r03i_pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-profile-pass"
run() { name=$1; shift
  envs=""; while [ $# -gt 0 ] && [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done
  env $envs $B "$@" > gpurun_out/r03i_$name.log 2>&1
  grep -a '"metric"' gpurun_out/r03i_$name.log | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('$name: ms_per_step %.2f %s residual %.2e factor_check %.2e replayed %s host_sched %.3f' % (l['ms_per_step'], l['step_ms'], l['residual'], l['factor_check'], l.get('static_schedule_replayed'), l['host_sched_s_last_step']))"
  grep -av metric gpurun_out/r03i_$name.log | tail -2
}
run fem112_replay --steps 5 --warmup 2
run fem112_noreplay PANGULU_AMD_REPLAY=0 --steps 5 --warmup 2
run shell_replay --workload shell --steps 20 --warmup 3
run shell_noreplay PANGULU_AMD_REPLAY=0 --workload shell --steps 20 --warmup 3
run shell_replay_oldkernel PANGULU_HIP_TILES_STAGES=0 PANGULU_HIP_FRONT_STAGES=0 --workload shell --steps 20 --warmup 3
run fem80_replay --size 80 --steps 8 --warmup 2
run fem80_noreplay PANGULU_AMD_REPLAY=0 --size 80 --steps 8 --warmup 2
run poisson96_replay --workload poisson --size 96 --steps 5 --warmup 2
