#!/bin/bash
# round 3, fifth GPU job: lazy updates (queues accumulate until the destination's own panel task) against the per-level look-ahead flush
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="timeout 900 python bench.py --no-cpu-baseline"
run() { # name, env...
  name=$1; shift
  env "$@" PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/r03e_launch_log_$name.txt $B --steps 3 --warmup 1 > gpurun_out/r03e_$name.log 2>&1
  grep -a '"metric"' gpurun_out/r03e_$name.log | python -c "
import sys,json
l=json.loads(sys.stdin.read()); k=l['kernels']; sd=k['ssssm_dense_mfma']
print('$name: ms_per_step %.1f (%s) residual %.2e factor_check %.2e batches %d; update kernel %.1f ms in %d launches, executed %.1f TF/s; trsm %.1f getrf %.1f densify %.1f sparsify %.1f' % (l['ms_per_step'], l['step_ms'], l['residual'], l['factor_check'], l['batches_per_step'], sd['ms'], sd['launches'], l['roofline']['mfma_executed_tflops'], k['tstrf']['ms'], k['getrf']['ms'], k['densify']['ms'], k['sparsify']['ms']))"
  python tools/launch_log_summary.py gpurun_out/r03e_launch_log_$name.txt | sed -n '1p;7,9p'
}
run default X=1
run lazy PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0
run lazy_small0 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_SMALL_LAUNCH_TASKS=0
run lazy_small0_chunk16 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_SMALL_LAUNCH_TASKS=0 PANGULU_HIP_GROUP_CHUNK=16
run lazy_small0_chunk4 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_SMALL_LAUNCH_TASKS=0 PANGULU_HIP_GROUP_CHUNK=4
run lazy_small256 PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_SMALL_LAUNCH_TASKS=256
run default_small0 PANGULU_HIP_SMALL_LAUNCH_TASKS=0
