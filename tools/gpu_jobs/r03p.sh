#!/bin/bash
# how many workgroups of the update kernels does a CU hold?  time of the first 256 / 512 / 1024 workgroups of a launch
cd /root/repo/tools/microbench
{
for k in 1 4 8; do
 for cap in 256 512 768 1024; do
  echo "=== k = $k, first $cap workgroups ==="
  GRID_CAP=$cap timeout 120 ./front_gemm.bin 32 8 -$k 2>&1 | grep -E "^time" | tail -5 | grep -E "round-2|1 dest"
 done
done
} > /root/repo/gpurun_out/r03p_residency.log 2>&1
cat /root/repo/gpurun_out/r03p_residency.log
