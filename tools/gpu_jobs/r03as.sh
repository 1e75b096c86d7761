#!/bin/bash
# queue length on the k-d ordered pattern (the dense-front kernel carries more of the work now and likes long queues)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  env "$@" timeout 900 python bench.py --gpu-worker --workload fem27 --size 112 --steps 3 --warmup 1 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-70s %.1f ms  residual %.2e' % ('$*', d['ms_per_step'], d['residual']))"
}
{
run PANGULU_AMD_X=0
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 PANGULU_HIP_GROUP_CHUNK=16
run PANGULU_HIP_GROUP_CHUNK=16
run PANGULU_AMD_LOOKAHEAD_MAX_GETRF=4
} 2>&1 | tee gpurun_out/r03as_queue_length_kd.log
