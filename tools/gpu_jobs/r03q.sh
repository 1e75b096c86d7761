#!/bin/bash
# per-workgroup overhead of the update kernels: the same number of slab steps as 1, 2, 4, 8 updates per destination
cd /root/repo/tools/microbench
{
for k in 8 5; do
 for pq in "64 1" "45 2" "32 4" "23 8"; do
  set -- $pq
  echo "=== k = $k, $1 x $1 destinations, $2 update(s) each ==="
  timeout 120 ./front_gemm.bin $1 $2 -$k 2>&1 | grep -E "^front|^time" | tail -6 | grep -E "^front|round-2|1 dest"
 done
done
} > /root/repo/gpurun_out/r03q_per_wg_overhead.log 2>&1
cat /root/repo/gpurun_out/r03q_per_wg_overhead.log
