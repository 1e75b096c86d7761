#!/bin/bash
# round 3, second GPU job: f64 MFMA ceiling (fixed), init phase times, per-launch logs of the update kernel, background look-ahead A/B,
# multi-rank + bench tests with the launcher thread / supervisor, one rank through the multi-rank loop
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/microbench/mfma_f64_peak.bin > gpurun_out/r03b_mfma_peak.log 2>&1
cat gpurun_out/r03b_mfma_peak.log
B="timeout 900 python bench.py --no-cpu-baseline"
PANGULU_AMD_TRACE=1 PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/r03b_launch_log_fem80.txt $B --size 80 --steps 3 --warmup 1 > gpurun_out/r03b_fem80.log 2>&1
grep -a "trace\] init" gpurun_out/r03b_fem80.log; grep -a '"metric"' gpurun_out/r03b_fem80.log | cut -c1-260
python tools/launch_log_summary.py gpurun_out/r03b_launch_log_fem80.txt > gpurun_out/r03b_launch_summary_fem80.txt 2>&1; cat gpurun_out/r03b_launch_summary_fem80.txt
PANGULU_AMD_TRACE=1 PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/r03b_launch_log_fem112.txt $B --steps 4 --warmup 1 > gpurun_out/r03b_fem112_bg1.log 2>&1
grep -a "trace\] init" gpurun_out/r03b_fem112_bg1.log; grep -a '"metric"' gpurun_out/r03b_fem112_bg1.log | cut -c1-330
python tools/launch_log_summary.py gpurun_out/r03b_launch_log_fem112.txt > gpurun_out/r03b_launch_summary_fem112.txt 2>&1; cat gpurun_out/r03b_launch_summary_fem112.txt
PANGULU_HIP_BACKGROUND_UPDATES=0 PANGULU_AMD_PANEL_FIRST=0 $B --steps 4 --warmup 1 --no-profile-pass > gpurun_out/r03b_fem112_bg0.log 2>&1
grep -a '"metric"' gpurun_out/r03b_fem112_bg0.log | cut -c1-330
$B --workload shell --steps 8 --warmup 2 --no-profile-pass > gpurun_out/r03b_shell_bg1.log 2>&1
grep -a '"metric"' gpurun_out/r03b_shell_bg1.log | cut -c1-330
PANGULU_HIP_BACKGROUND_UPDATES=0 PANGULU_AMD_PANEL_FIRST=0 $B --workload shell --steps 8 --warmup 2 --no-profile-pass > gpurun_out/r03b_shell_bg0.log 2>&1
grep -a '"metric"' gpurun_out/r03b_shell_bg0.log | cut -c1-330
PANGULU_AMD_FORCE_MULTI_LOOP=1 PANGULU_AMD_TRACE=1 $B --workload shell --steps 8 --warmup 2 --no-profile-pass > gpurun_out/r03b_shell_multiloop.log 2>&1
grep -a '"metric"' gpurun_out/r03b_shell_multiloop.log | cut -c1-330; grep -a "scheduler loop" gpurun_out/r03b_shell_multiloop.log | tail -1
( time timeout 1200 python -m pytest tests/test_gpu_smoke_bench.py tests/test_multirank.py tests/test_gpu_env_switches.py -m gpu -x -q ) > gpurun_out/r03b_pytest.log 2>&1
tail -15 gpurun_out/r03b_pytest.log
