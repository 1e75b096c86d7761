#!/bin/bash
# round 3, eighth GPU job: checkpoint -- whole GPU suite, default bench line (with CPU leg), ldoor-class line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/r03h_pytest.log 2>&1
tail -12 gpurun_out/r03h_pytest.log
( time timeout 1500 python bench.py ) > gpurun_out/r03h_bench_default.log 2> gpurun_out/r03h_bench_default.err
tail -c 1500 gpurun_out/r03h_bench_default.log; tail -3 gpurun_out/r03h_bench_default.err
( time timeout 900 python bench.py --workload shell --steps 20 --warmup 3 ) > gpurun_out/r03h_bench_shell.log 2> gpurun_out/r03h_bench_shell.err
cut -c1-400 gpurun_out/r03h_bench_shell.log; tail -3 gpurun_out/r03h_bench_shell.err
