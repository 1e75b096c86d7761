#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04v_cr64_diag6.log
: > $OUT
run() { echo "== $*" | tee -a $OUT; env "${@:3}" timeout 900 python tools/cr64_diag.py $1 $2 2>&1 | tail -1 | tee -a $OUT; }
run 48 128 DIAG_VTYPE=r64 DIAG_PERMILLE=100
run 64 128 DIAG_VTYPE=r64 DIAG_PERMILLE=50
run 48 128 DIAG_PERMILLE=100
run 48 128 DIAG_PERMILLE=300
run 64 128 DIAG_PERMILLE=20
run 64 128 DIAG_PERMILLE=100
run 76 128 DIAG_PERMILLE=10
run 80 256 DIAG_PERMILLE=10
run 48 128 DIAG_PERMILLE=100 PANGULU_AMD_REPLAY=0
run 48 128 DIAG_PERMILLE=100 DIAG_RESETS=2
