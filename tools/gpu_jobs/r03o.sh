#!/bin/bash
# LDS counters instead of the per-step barrier in the tilesv kernel: checks, step cost, probes
cd /root/repo/tools/microbench
{
timeout 120 ./front_gemm.bin 8 2 100 2>&1 | grep -E "^check"
for k in 1 2 3 4 5 6 8; do
  echo "=== k = $k ==="
  timeout 120 ./front_gemm_probe.bin 32 8 -$k 2>&1 | grep -E "^front|^time|^probe" | tail -4
  timeout 120 ./front_gemm.bin 32 8 -$k 2>&1 | grep -E "^time" | tail -5
done
echo "=== 45 % random ranges ==="
timeout 120 ./front_gemm.bin 32 8 45 2>&1 | grep -E "^front|^time" | tail -6
echo "=== full, 4 updates each ==="
timeout 120 ./front_gemm.bin 40 4 100 2>&1 | grep -E "^front|^time" | tail -7
} > /root/repo/gpurun_out/r03o_tilesv.log 2>&1
cat /root/repo/gpurun_out/r03o_tilesv.log
