#!/bin/bash
# round 3, twelfth GPU job: tiles kernel with the destination preloaded, front launch threshold, new thresholds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "40 1 100" "40 1 45"; do
  timeout 600 tools/microbench/front_gemm.bin $cfg > gpurun_out/r03l_front_gemm_$(echo $cfg | tr ' ' '_').log 2>&1
  grep -E "^check|^front|^time" gpurun_out/r03l_front_gemm_$(echo $cfg | tr ' ' '_').log | grep -v "ok$" | head -12
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_update_values.py -m gpu -x -q ) > gpurun_out/r03l_pytest.log 2>&1; tail -4 gpurun_out/r03l_pytest.log
B="timeout 900 python bench.py --no-cpu-baseline --no-profile-pass"
run() { name=$1; shift
  envs=""; while [ $# -gt 0 ] && [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done
  env $envs $B "$@" > gpurun_out/r03l_$name.log 2>&1
  grep -a '"metric"' gpurun_out/r03l_$name.log | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('$name: ms_per_step %.2f (min %.2f) residual %.2e replayed %s' % (l['ms_per_step'], min(l['step_ms']), l['residual'], l.get('static_schedule_replayed')))"
}
F="--steps 4 --warmup 2"
run fem_default $F
run fem_front1 PANGULU_HIP_FRONT_STAGES=1 $F
run fem_minwgs512 PANGULU_HIP_FRONT_MIN_WGS=512 $F
run fem_minwgs8192 PANGULU_HIP_FRONT_MIN_WGS=8192 $F
run fem_dense1 PANGULU_HIP_DENSE_PERMILLE=1 $F
run fem_dense3 PANGULU_HIP_DENSE_PERMILLE=3 $F
run fem_la32 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=32 $F
S="--workload shell --steps 10 --warmup 2"
run shell_default $S
run shell_front1 PANGULU_HIP_FRONT_STAGES=1 $S
run shell_dense1 PANGULU_HIP_DENSE_PERMILLE=1 $S
run shell_dense3 PANGULU_HIP_DENSE_PERMILLE=3 $S
run shell_trsm2 PANGULU_HIP_TRSM_DENSE_PERMILLE=2 $S
run shell_la32 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=32 $S
run poisson96 --workload poisson --size 96 --steps 4 --warmup 2
run fem80 --size 80 --steps 6 --warmup 2
