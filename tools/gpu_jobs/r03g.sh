#!/bin/bash
# round 3, seventh GPU job: where the non-update time goes with the current build: kernel traces (default, lazy), launch chunking
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="timeout 900 python bench.py --no-cpu-baseline --no-profile-pass"
run() { name=$1; shift
  env "$@" $B --steps 4 --warmup 1 > gpurun_out/r03g_$name.log 2>&1
  grep -a '"metric"' gpurun_out/r03g_$name.log | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('$name: ms_per_step %.1f %s residual %.2e batches %d host_sched %.3f' % (l['ms_per_step'], l['step_ms'], l['residual'], l['batches_per_step'], l['host_sched_s_last_step']))"
}
run default X=1
run chunk1024 PANGULU_HIP_LAUNCH_CHUNK=1024
run chunk4096 PANGULU_HIP_LAUNCH_CHUNK=4096
run lazy PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0
run lazy_chunk2048 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_LAUNCH_CHUNK=2048
run bg0 PANGULU_HIP_BACKGROUND_UPDATES=0 PANGULU_AMD_PANEL_FIRST=0
run old_kernel PANGULU_HIP_TILES_STAGES=0 PANGULU_HIP_FRONT_STAGES=0
cd /tmp && export TMPDIR=/tmp
for v in default lazy; do
  E=""; [ $v = lazy ] && export PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03g_$v -o runc -- python3 $R/bench.py --no-cpu-baseline --no-profile-pass --steps 2 --warmup 1 2>&1 | grep -a metric | cut -c1-150
  T=$(find $R/gpurun_out/prof_r03g_$v -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/critical_path.py $T > $R/gpurun_out/r03g_critical_path_$v.md 2>&1
  cat $R/gpurun_out/r03g_critical_path_$v.md
  find $R/gpurun_out/prof_r03g_$v -name "*kernel_trace.csv" -delete
done
