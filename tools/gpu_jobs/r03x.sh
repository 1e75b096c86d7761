#!/bin/bash
# longer update queues per tile with the current kernels (replayed): lazy updates, larger group chunks
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload fem27 --size 112 --steps 3 --warmup 1 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['kernels']['ssssm_dense_mfma']
print('%-60s %.1f ms  residual %.2e  update kernels %.1f ms in %d launches, workgroups %s' % ('$*', d['ms_per_step'], d['residual'], k['ms'], k['launches'], k.get('workgroups')))"
}
{
run PANGULU_AMD_X=0
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0
run PANGULU_HIP_GROUP_CHUNK=16
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_GROUP_CHUNK=16
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=8
} 2>&1 | tee gpurun_out/r03x_queue_length.log
