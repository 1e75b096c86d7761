#!/bin/bash
# LDS counters instead of the per-step barrier in the tilesv kernel (ssssm_tilesv_f64_kernel<1>): checks, then step cost against fill
cd /root/repo/tools/microbench
{
timeout 120 ./front_gemm.bin 8 2 100 2>&1 | grep -E "^check" | grep -E "issue behind"
timeout 120 ./front_gemm.bin 8 20 45 2>&1 | grep -E "^check" | grep -E "issue behind"
for k in 2 4 5 6 8; do
  echo "=== k = $k ==="
  timeout 120 ./front_gemm.bin 32 8 -$k 2>&1 | grep -E "^time" | grep -E "issue behind" | tail -4
done
echo "=== 45 % random ranges, 8 and 2 updates each ==="
timeout 120 ./front_gemm.bin 32 8 45 2>&1 | grep -E "^time" | grep -E "issue behind" | tail -4
timeout 120 ./front_gemm.bin 45 2 45 2>&1 | grep -E "^time" | grep -E "issue behind" | tail -4
} > /root/repo/gpurun_out/r03ag_lds_counters.log 2>&1
cat /root/repo/gpurun_out/r03ag_lds_counters.log
