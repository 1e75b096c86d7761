#!/bin/bash
# stream kernel (work items drawn by ticket, next unit prepared by a clerk wavefront) against the tilesv kernel
cd /root/repo/tools/microbench
{
timeout 120 ./front_gemm.bin 8 2 100 2>&1 | grep -E "^check" | grep -E "stream|issue behind"
timeout 120 ./front_gemm.bin 8 20 100 2>&1 | grep -E "^check" | grep -E "stream|issue behind"
for k in 8 5 3; do
 for pq in "64 1" "45 2" "32 4"; do
  set -- $pq
  echo "=== k = $k, $1 x $1 destinations, $2 update(s) each ==="
  timeout 120 ./front_gemm.bin $1 $2 -$k 2>&1 | grep -E "^time" | tail -8
 done
done
echo "=== 45 % random ranges, 1 and 2 updates each ==="
timeout 120 ./front_gemm.bin 64 1 45 2>&1 | grep -E "^front|^time" | tail -9
timeout 120 ./front_gemm.bin 45 2 45 2>&1 | grep -E "^front|^time" | tail -9
} > /root/repo/gpurun_out/r03s_stream.log 2>&1
cat /root/repo/gpurun_out/r03s_stream.log
