#!/bin/bash
# round 3, first GPU job: GPU test suite, default bench line, MFMA f64 ceiling, update-kernel phase stamps, kernel trace of the Serena-class run
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03a_pytest.log 2>&1
tail -5 gpurun_out/r03a_pytest.log
tools/microbench/mfma_f64_peak.bin > gpurun_out/r03a_mfma_peak.log 2>&1
cat gpurun_out/r03a_mfma_peak.log
( time timeout 1500 python bench.py ) > gpurun_out/r03a_bench_default.log 2> gpurun_out/r03a_bench_default.err
tail -c 3000 gpurun_out/r03a_bench_default.log; tail -5 gpurun_out/r03a_bench_default.err
PANGULU_HIP_DEBUG_SSSSM=1 timeout 600 python bench.py --size 80 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/r03a_stamps_fem80.log 2>&1
grep -a "stamps" gpurun_out/r03a_stamps_fem80.log | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03a -o runc -- python3 $R/bench.py --no-cpu-baseline --no-profile-pass --steps 2 --warmup 1 2>&1 | grep -a metric | cut -c1-200
cd $R
T=$(find gpurun_out/prof_r03a -name "*kernel_trace.csv" | head -1)
python tools/launch_size_histogram.py $T > gpurun_out/r03a_launch_hist_fem112.txt 2>&1
python tools/critical_path.py $T > gpurun_out/r03a_critical_path_fem112.md 2>&1
# per-launch table of the update kernel (workgroups, us) for the design of the dense fast path
python - "$T" > gpurun_out/r03a_ssssm_launches_fem112.csv <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'ssssm_dense' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print("start_ns,wgs,us")
for r in rows:
    wg=int(r.get('Grid_Size_X') or r.get('Grid_Size'))//int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
    print("%s,%d,%.1f"%(r['Start_Timestamp'],wg,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
gzip -9 -c $T > gpurun_out/r03a_kernel_trace_fem112.csv.gz
find gpurun_out/prof_r03a -name "*kernel_trace.csv" -delete
ls -la gpurun_out | tail -15
