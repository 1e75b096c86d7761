#!/bin/bash
# The one reproducible profiling recipe (run on the GPU box from the repo root, e.g. through gpurun):
#   tools/profile_recipe.sh <tag> [extra bench.py flags]
# writes rocprofv3 kernel stats + the PMC passes (each in its own run: --pmc never together with other trace domains)
# under gpurun_out/prof_<tag>*, summarises them into gpurun_out/<tag>_table.md, gpurun_out/<tag>_critical_path.md and merges the
# per-launch HBM bytes of every kernel into gpurun_out/hbm_traffic.json under the workload's key (bench.workload_key).
# Copy what you want judged into profiles/ (bench.py quotes profiles/hbm_traffic.json only for the build and workload it profiled).
# With one rank the first factorisation of the run records the launch schedule and the others replay it: same kernels.
# PANGULU_HIP_FRONT_FORK=0 in every pass (round 6): the timed configuration runs the dense-front launch of an update call on a stream
# of its own beside the general launch (0.7 % faster); in a kernel trace the two then overlap and their durations are no longer additive.
# The passes here keep them on one stream -- the configuration of bench.py's own profile pass, which is where the line's per-kernel
# times and its roofline come from.
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
export PANGULU_HIP_FRONT_FORK=0
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpu-worker --no-profile-pass --no-secondary --no-sched-steps --steps 2 --warmup 1 $*"
O=$R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o runc -- $B 2>&1 | grep metric | cut -c1-160
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -o runc -- $B 2>&1 | grep metric | cut -c1-120
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -o runc -- $B 2>&1 | grep metric | cut -c1-120
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d ${O}_sq -o runc -- $B 2>&1 | grep metric | cut -c1-120
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${O}_mfma -o runc -- $B 2>&1 | grep -i metric | cut -c1-120
cd $R
export WORKLOAD_KEY="$(python3 -c "import sys; sys.argv=['bench.py']+'$*'.split(); import bench; print(bench.workload_key(bench.parse_args()))")"
PROFILE_NAME=profiles/${TAG}.md python3 tools/summarize_rocprof.py $O ${O}_fetch ${O}_write ${O}_sq ${O}_mfma --json gpurun_out/hbm_traffic.json > gpurun_out/${TAG}_table.md
python3 tools/critical_path.py $(find $O -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_critical_path.md
find gpurun_out/prof_${TAG}* -name "*kernel_trace.csv" -delete
