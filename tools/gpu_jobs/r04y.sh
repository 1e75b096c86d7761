#!/bin/bash
# kernel statistics of the complex Poisson class (poisson3d(80) CR64, nb = 128; three factorisations)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0 DIAG_RESETS=2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04y2 -o runc -- python3 $R/tools/cr64_diag.py 80 128 2>&1 | tail -2
cd $R
f=$(find gpurun_out/prof_r04y2 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r04y2_cr64_poisson80_kernel_stats.csv
head -14 $f | cut -c1-200
python3 tools/critical_path.py $(find gpurun_out/prof_r04y2 -name "*kernel_trace.csv" | head -1) > gpurun_out/r04y2_cr64_critical_path.md
head -16 gpurun_out/r04y2_cr64_critical_path.md | cut -c1-200
find gpurun_out/prof_r04y2 -name "*kernel_trace.csv" -delete
