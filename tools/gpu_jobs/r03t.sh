#!/bin/bash
# static wave priorities against the lockstep of a CU's two workgroups (tilesv kernel; modes: 0 none, 1 by generation of the
# workgroup, 2 hashed 0..3, 3 waves 4-7 of every workgroup, 4 generation x wave half)
cd /root/repo/tools/microbench
{
for k in 8 6 5 4 3 2; do
  echo "=== k = $k, 32 x 32 destinations, 8 updates each ==="
  timeout 120 ./front_gemm.bin 32 8 -$k 2>&1 | grep -E "^time" | tail -5
done
echo "=== 45 % random ranges ==="
timeout 120 ./front_gemm.bin 32 8 45 2>&1 | grep -E "^time" | tail -5
} > /root/repo/gpurun_out/r03t_priorities.log 2>&1
cat /root/repo/gpurun_out/r03t_priorities.log
