#!/bin/bash
# round 4: where the ldoor-class factorisation (shell(398), 37 ms) spends its time now: kernel trace -> exclusive time per class, per-kernel stats
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$R/gpurun_out/prof_r04j_shell
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o runc -- python3 $R/bench.py --gpu-worker --workload shell --no-profile-pass --no-secondary --steps 3 --warmup 1 2>&1 | grep metric | cut -c1-200
cd $R
python3 tools/critical_path.py $(find $O -name "*kernel_trace.csv" | head -1) > gpurun_out/r04j_shell398_critical_path.md
cat gpurun_out/r04j_shell398_critical_path.md | cut -c1-220
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r04j_shell/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
out = open('gpurun_out/r04j_shell398_kernel_stats.txt', 'w')
for r in rows[:14]:
    line = "%-70s calls %6s total %10.1f us avg %8.1f us max %8.1f" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3)
    print(line); out.write(line + "\n")
PY
python3 tools/launch_size_histogram.py $(find $O -name "*kernel_trace.csv" | head -1) 2>/dev/null | head -40 > gpurun_out/r04j_shell398_launch_sizes.txt; head -30 gpurun_out/r04j_shell398_launch_sizes.txt | cut -c1-200
find gpurun_out/prof_r04j_shell -name "*kernel_trace.csv" -delete
