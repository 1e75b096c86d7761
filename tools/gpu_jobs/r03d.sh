#!/bin/bash
# round 3, fourth GPU job: the LDS-DMA tiles kernel (general update kernel) stand-alone on full and partly filled fronts, its parity
# tests, and inside the Serena-class factorisation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "40 1 100" "40 1 45" "24 3 45" "40 1 25"; do
  timeout 600 tools/microbench/front_gemm.bin $cfg > gpurun_out/r03d_front_gemm_$(echo $cfg | tr ' ' '_').log 2>&1
  grep -E "^check|^front|^time" gpurun_out/r03d_front_gemm_$(echo $cfg | tr ' ' '_').log | grep -v "ok$" | head -14
done
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_front or general_update" ) > gpurun_out/r03d_pytest.log 2>&1; tail -5 gpurun_out/r03d_pytest.log
B="timeout 900 python bench.py --no-cpu-baseline"
for cfg in "2 2" "0 2" "0 0"; do
set -- $cfg
PANGULU_HIP_FRONT_STAGES=$1 PANGULU_HIP_TILES_STAGES=$2 PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/r03d_launch_log_fem112_f$1_t$2.txt $B --steps 3 --warmup 1 > gpurun_out/r03d_fem112_f$1_t$2.log 2>&1
grep -a '"metric"' gpurun_out/r03d_fem112_f$1_t$2.log | python -c "
import sys,json
l=json.loads(sys.stdin.read()); k=l['kernels']['ssssm_dense_mfma']
print('front stages $1 tiles stages $2: ms_per_step %.1f residual %.2e factor_check %.2e; update kernel %.1f ms, %s, executed %.1f TF/s' % (l['ms_per_step'], l['residual'], l['factor_check'], k['ms'], k.get('workgroups'), l['roofline']['mfma_executed_tflops']))"
python tools/launch_log_summary.py gpurun_out/r03d_launch_log_fem112_f$1_t$2.txt | head -12
done
