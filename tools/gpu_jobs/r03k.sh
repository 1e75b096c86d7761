#!/bin/bash
# round 3, eleventh GPU job: option sweep on replayed runs (timings reproducible to 0.1 %)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="timeout 900 python bench.py --no-cpu-baseline --no-profile-pass"
run() { name=$1; shift
  envs=""; while [ $# -gt 0 ] && [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done
  env $envs $B "$@" > gpurun_out/r03k_$name.log 2>&1
  grep -a '"metric"' gpurun_out/r03k_$name.log | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('$name: ms_per_step %.2f (min %.2f) residual %.2e replayed %s' % (l['ms_per_step'], min(l['step_ms']), l['residual'], l.get('static_schedule_replayed')))"
}
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_front or general_update" ) > gpurun_out/r03k_pytest.log 2>&1; tail -4 gpurun_out/r03k_pytest.log
F="--steps 4 --warmup 2"
run fem_default $F
run fem_front2 PANGULU_HIP_FRONT_STAGES=2 $F
run fem_front0 PANGULU_HIP_FRONT_STAGES=0 $F
run fem_tiles_unit8 PANGULU_HIP_TILES_UNIT=8 PANGULU_HIP_FRONT_UNIT=8 $F
run fem_tiles_unit4 PANGULU_HIP_TILES_UNIT=4 PANGULU_HIP_FRONT_UNIT=4 $F
run fem_dense2 PANGULU_HIP_DENSE_PERMILLE=2 $F
run fem_dense10 PANGULU_HIP_DENSE_PERMILLE=10 $F
run fem_chunk4 PANGULU_HIP_GROUP_CHUNK=4 $F
run fem_chunk12 PANGULU_HIP_GROUP_CHUNK=12 $F
run fem_small512 PANGULU_HIP_SMALL_LAUNCH_TASKS=512 $F
run fem_la16 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=16 $F
run fem_la1024 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=1024 $F
run fem_trsm5 PANGULU_HIP_TRSM_DENSE_PERMILLE=5 $F
run fem_trsm30 PANGULU_HIP_TRSM_DENSE_PERMILLE=30 $F
S="--workload shell --steps 10 --warmup 2"
run shell_default $S
run shell_front2 PANGULU_HIP_FRONT_STAGES=2 $S
run shell_tiles_unit8 PANGULU_HIP_TILES_UNIT=8 PANGULU_HIP_FRONT_UNIT=8 $S
run shell_dense2 PANGULU_HIP_DENSE_PERMILLE=2 $S
run shell_dense10 PANGULU_HIP_DENSE_PERMILLE=10 $S
run shell_la16 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=16 $S
run shell_la1024 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=1024 $S
run shell_trsm5 PANGULU_HIP_TRSM_DENSE_PERMILLE=5 $S
run shell_trsm30 PANGULU_HIP_TRSM_DENSE_PERMILLE=30 $S
run shell_chunk4 PANGULU_HIP_GROUP_CHUNK=4 $S
