#!/bin/bash
# round 4 experiment: dense-front kernel on 128 x 64 tiles, four wavefronts per workgroup, three workgroups per CU (tools/experiments/front_n64.h)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( cd tools/microbench && for q in 1 4; do timeout 300 ./front_gemm.bin 40 $q 100; done ) > gpurun_out/r04f_front_n64.log 2>&1
grep -E "check.*(K = 32|128 x 64)|time" gpurun_out/r04f_front_n64.log | cut -c1-200
