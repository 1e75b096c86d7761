#!/bin/bash
# tilesv kernel inside the factorisation: parity tests of the update kernels, then A/B on one box (TILES_STAGES 2 = new, 1 = previous)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py tests/test_update_values.py -m gpu -q -x ) > gpurun_out/r03r_pytest.log 2>&1
tail -5 gpurun_out/r03r_pytest.log
for st in 2 1 2 1; do
  PANGULU_HIP_TILES_STAGES=$st timeout 900 python bench.py --gpu-worker --workload fem27 --size 112 --steps 4 --warmup 1 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('fem27(112) TILES_STAGES=$st: %.1f ms  residual %.2e  update kernels %.1f ms  roofline %.3f' % (d['ms_per_step'], d['residual'], d['kernels']['ssssm_dense_mfma']['ms'], d['roofline']['frac']))"
done 2>&1 | tee gpurun_out/r03r_ab_fem27.log
for st in 2 1 2 1; do
  PANGULU_HIP_TILES_STAGES=$st timeout 900 python bench.py --gpu-worker --workload shell --steps 20 --warmup 3 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('shell(398) TILES_STAGES=$st: %.2f ms  residual %.2e' % (d['ms_per_step'], d['residual']))"
done 2>&1 | tee gpurun_out/r03r_ab_shell.log
