#!/bin/bash
# round 3, third GPU job: dense-front kernel (LDS-DMA pipeline) stand-alone and inside the factorisation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 tools/microbench/front_gemm.bin 40 1 > gpurun_out/r03c_front_gemm_q1.log 2>&1; cat gpurun_out/r03c_front_gemm_q1.log
timeout 600 tools/microbench/front_gemm.bin 24 4 > gpurun_out/r03c_front_gemm_q4.log 2>&1; grep time gpurun_out/r03c_front_gemm_q4.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_front or dense_blocks or thresholds" ) > gpurun_out/r03c_pytest.log 2>&1; tail -5 gpurun_out/r03c_pytest.log
B="timeout 900 python bench.py --no-cpu-baseline"
for st in 3 0; do
PANGULU_HIP_FRONT_STAGES=$st PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/r03c_launch_log_fem112_front$st.txt $B --steps 3 --warmup 1 > gpurun_out/r03c_fem112_front$st.log 2>&1
grep -a '"metric"' gpurun_out/r03c_fem112_front$st.log | python -c "
import sys,json
l=json.loads(sys.stdin.read()); k=l['kernels']['ssssm_dense_mfma']
print('front stages $st: ms_per_step %.1f residual %.2e factor_check %.2e; update kernel %.1f ms, %s, executed %.1f TF/s' % (l['ms_per_step'], l['residual'], l['factor_check'], k['ms'], k.get('workgroups'), l['roofline']['mfma_executed_tflops']))"
python tools/launch_log_summary.py gpurun_out/r03c_launch_log_fem112_front$st.txt | head -12
done
