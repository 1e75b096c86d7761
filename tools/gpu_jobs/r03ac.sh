#!/bin/bash
# kernel durations of the CR64 run with the complex panels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03ac -o runc -- python3 $R/tools/bench_types.py 48 cr64 2>&1 | grep poisson | cut -c1-160

python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r03ac/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-70s calls %5s  total %9.1f us  avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
find gpurun_out/prof_r03ac -name "*kernel_trace.csv" -delete
