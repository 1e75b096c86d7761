#!/bin/bash
# complex TSTRF / GESSM on the matrix cores (ztrsm_direct_kernel, nb = 128): parity cases of the complex types, then A/B timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_parity.py -m gpu -x -q -k "cr64 or cr32 or 128 or c_" ) > gpurun_out/r04z_complex_tests.log 2>&1
tail -5 gpurun_out/r04z_complex_tests.log
OUT=gpurun_out/r04z_ztrsm_ab.log
: > $OUT
run() { echo "== $*" | tee -a $OUT; env "${@:3}" timeout 900 python tools/cr64_diag.py $1 $2 2>&1 | tail -1 | tee -a $OUT; }
for N in 48 64 80; do
run $N 128 DIAG_RESETS=2
run $N 128 DIAG_RESETS=2 PANGULU_HIP_ZTRSM_DIRECT=0
done
