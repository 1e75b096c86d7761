#!/bin/bash
# chase only for batches of at most N factorisations; kernel durations with rocprofv3
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {
  env "$@" timeout 600 python bench.py --gpu-worker --workload $W --steps $S --warmup 2 --no-profile-pass 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s %-52s %.2f ms  residual %.2e' % ('$W', '$*', d['ms_per_step'], d['residual']))"
}
{
W=shell; S=20
run PANGULU_HIP_CHASE=0
run PANGULU_HIP_CHASE=1 PANGULU_HIP_CHASE_MAX_GETRF=1
run PANGULU_HIP_CHASE=1 PANGULU_HIP_CHASE_MAX_GETRF=2
run PANGULU_HIP_CHASE=1 PANGULU_HIP_CHASE_MAX_GETRF=4
run PANGULU_HIP_CHASE=1 PANGULU_HIP_CHASE_MAX_GETRF=16
run PANGULU_HIP_CHASE=0
} 2>&1 | tee gpurun_out/r03z2_chase_max.log
cd /tmp && export TMPDIR=/tmp
PANGULU_HIP_CHASE=1 PANGULU_HIP_CHASE_MAX_GETRF=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03z2 -o runc -- python3 $R/bench.py --gpu-worker --no-profile-pass --steps 2 --warmup 1 --workload shell 2>&1 | grep metric | cut -c1-100
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r03z2/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print('%-60s calls %5s  total %9.1f us  avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
find gpurun_out/prof_r03z2 -name "*kernel_trace.csv" -delete
