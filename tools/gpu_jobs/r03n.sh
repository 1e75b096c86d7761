#!/bin/bash
# step cost of the general update kernel against the number of live products in a step (k x k of 64), with cycle probes
cd /root/repo/tools/microbench
for k in 1 2 3 4 5 6 8; do
  echo "=== k = $k ==="
  timeout 120 ./front_gemm_probe.bin 32 8 -$k 2>&1 | grep -E "^front|^time|^probe"
  timeout 120 ./front_gemm.bin 32 8 -$k 2>&1 | grep -E "^time" | head -5
done > /root/repo/gpurun_out/r03n_step_cost.log 2>&1
timeout 120 ./front_gemm_probe.bin 32 8 100 2>&1 | grep -E "^front|^time|^probe" >> /root/repo/gpurun_out/r03n_step_cost.log
timeout 120 ./front_gemm_probe.bin 32 8 45 2>&1 | grep -E "^front|^time|^probe" >> /root/repo/gpurun_out/r03n_step_cost.log
cat /root/repo/gpurun_out/r03n_step_cost.log
