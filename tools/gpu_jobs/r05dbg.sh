#!/bin/bash
# round 5: the 8-rank host-staged case that failed in the full suite, three times with the ring kernel and once without
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05dbg}
for i in 1 2 3; do
( timeout 600 python -m pytest "tests/test_multirank.py::test_default_map_on_the_gpu" -m gpu -q -x -k "8-shell_40x40-256-r64-host" ) > gpurun_out/${TAG}_ring1_$i.log 2>&1
tail -3 gpurun_out/${TAG}_ring1_$i.log
done
( PANGULU_HIP_TRSM_RING=0 timeout 600 python -m pytest "tests/test_multirank.py::test_default_map_on_the_gpu" -m gpu -q -x -k "8-shell_40x40-256-r64-host" ) > gpurun_out/${TAG}_ring0.log 2>&1
tail -3 gpurun_out/${TAG}_ring0.log
grep -h -B2 -A12 "rank 7 ---" gpurun_out/${TAG}_ring1_*.log | head -80
dmesg 2>/dev/null | tail -5
